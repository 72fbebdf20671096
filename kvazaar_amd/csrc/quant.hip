// quant.hip -- quant / dequant / coeff_abs_sum and the fused quantize_residual
// for gfx950.  Reference: src/strategies/generic/quant-generic.c (cited per
// kernel), transform.c:129-180 for the scaled QP and transform-skip.
//
// The encoder state the reference reads through `state` is flattened by the
// host into quant_consts (kvz_hip_quant_params in the C ABI).
#include "kvz_hip_internal.h"
#include "transform_core.h"

using namespace kvzhip;

struct quant_consts {
  int q_bits, add, flat_qc, signhide;       // quant
  const int32_t *qtable;                     // per-coefficient factors or nullptr (flat)
  int dq_mode;                               // 0 flat, 1 scaling list (shift > qp/6), 2 scaling list (clip + shl)
  int dq_shift, dq_add, dq_scale;            // mode 0: (q*scale + add) >> shift; mode 1: shift/add; mode 2: shl = dq_shift
  const int32_t *dqtable;
};

// transform.c:129-143
static int scaled_qp(int type, int qp)
{
  static const unsigned char chroma_scale[58] = {
     0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,
    33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };
  if (type == 0) return qp;
  int q = qp < 0 ? 0 : (qp > 57 ? 57 : qp);      // CLIP(-qp_offset, 57, qp) with qp_offset 0
  return chroma_scale[q];
}
static int log2i(int w) { int l = 0; while ((1 << l) < w) ++l; return l; }

// quant-generic.c:40-50 and :283-320
static bool make_consts(const kvz_hip_quant_params *p, int width, int type_q, int type_dq, quant_consts *c)
{
  static const int quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };     // scalinglist.c:66
  static const int inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                   // scalinglist.c:67
  if (width != 4 && width != 8 && width != 16 && width != 32) return false;
  const int log2_tr = log2i(width);
  const int transform_shift = 15 - 8 - log2_tr;
  {
    const int qps = scaled_qp(type_q, p->qp);
    if (qps < 0) return false;
    c->q_bits = 14 + qps / 6 + transform_shift;
    c->add = (p->slice_is_intra ? 171 : 85) << (c->q_bits - 9);
    c->flat_qc = quant_scales[qps % 6];
    c->signhide = p->signhide;
    c->qtable = (p->scaling_list && p->quant_coeff) ? p->quant_coeff : nullptr;
  }
  {
    const int qps = scaled_qp(type_dq, p->qp);
    int shift = 20 - 14 - transform_shift;
    if (p->scaling_list && p->dequant_coeff) {
      shift += 4;
      c->dqtable = p->dequant_coeff;
      if (shift > qps / 6) { c->dq_mode = 1; c->dq_shift = shift - qps / 6; c->dq_add = 1 << (c->dq_shift - 1); }
      else { c->dq_mode = 2; c->dq_shift = qps / 6 - shift; c->dq_add = 0; }
      c->dq_scale = 0;
    } else {
      c->dq_mode = 0; c->dqtable = nullptr;
      c->dq_scale = inv_quant_scales[qps % 6] << (qps / 6);
      c->dq_shift = shift; c->dq_add = 1 << (shift - 1);
    }
  }
  return true;
}

// quant-generic.c:55-67: unsigned level before sign/clip
__device__ __forceinline__ int quant_level(int c, int qc, const quant_consts &k)
{
  const int a = c < 0 ? -c : c;
  // flat quantisation: |c| <= 2^15 and quant_scales < 2^15, so the product is a full-rate 24-bit multiply and the sum
  // stays below 2^31 (add < 2^26); only scaling lists need the reference's 64-bit product (v_mul_lo_u32 and the 64-bit
  // multiply-add run at a quarter of the rate)
  if (!k.qtable) return (int)((__umul24((unsigned)a, (unsigned)qc) + (unsigned)k.add) >> k.q_bits);
  return (int)(((long long)a * qc + k.add) >> k.q_bits);
}
__device__ __forceinline__ int quant_one(int c, int qc, const quant_consts &k)
{
  int level = quant_level(c, qc, k);
  level = c < 0 ? -level : level;
  return clip16(level);
}
// quant-generic.c:290-320
__device__ __forceinline__ int dequant_one(int q, int n, const quant_consts &k)
{
  if (k.dq_mode == 0) return clip16((int)((unsigned)__mul24(q, k.dq_scale) + (unsigned)k.dq_add) >> k.dq_shift);   // |q| <= 2^15, scale <= 72 << 8
  const int d = k.dqtable[n];
  if (k.dq_mode == 1) return clip16((q * d + k.dq_add) >> k.dq_shift);
  int v = clip16(q * d);
  return clip16((int)((unsigned)v << k.dq_shift));
}

// ---- elementwise kernels: 8 coefficients (16 bytes) per lane ----
__global__ __launch_bounds__(256) void quant_kernel(const i16 *__restrict__ coef, i16 *__restrict__ q_coef,
                                                    size_t total, int block_elems, quant_consts k)
{
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (size_t i = tid * 8; i < total; i += nthreads * 8) {
    union { uint4 v; i16 s[8]; } in, out;
    in.v = ld_stream_u4(coef + i);
    const int n0 = (int)(i % (size_t)block_elems);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int qc = k.qtable ? k.qtable[n0 + j] : k.flat_qc;
      out.s[j] = (i16)quant_one(in.s[j], qc, k);
    }
    st_stream_u4(q_coef + i, out.v);
  }
}

__global__ __launch_bounds__(256) void dequant_kernel(const i16 *__restrict__ q_coef, i16 *__restrict__ coef,
                                                      size_t total, int block_elems, quant_consts k)
{
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (size_t i = tid * 8; i < total; i += nthreads * 8) {
    union { uint4 v; i16 s[8]; } in, out;
    in.v = ld_stream_u4(q_coef + i);
    const int n0 = (int)(i % (size_t)block_elems);
#pragma unroll
    for (int j = 0; j < 8; ++j) out.s[j] = (i16)dequant_one(in.s[j], n0 + j, k);
    st_stream_u4(coef + i, out.v);
  }
}

// ---- sign bit hiding (quant-generic.c:69-162) on one block; coef/q_coef may be
// global or LDS pointers.  Sequential per block, exactly the reference's control
// flow (including `abssum` being a signed sum and `cur_change` persisting across
// iterations).  Only reached with --signhide (off at preset medium). ----
// position of scan index `idx` for (scan_idx, log2 size): kvz_g_sig_last_scan
// (tables.c): 4x4 coefficient groups, group order and in-group order both follow
// the pattern (0 up-right diagonal, 1 horizontal, 2 vertical).
__device__ __forceinline__ int pattern_pos4(int scan_idx, int i)     // i in 0..15 -> y*4 + x inside a 4x4
{
  if (scan_idx == 1) return i;
  if (scan_idx == 2) return ((i & 3) << 2) | (i >> 2);
  const unsigned char t[16] = { 0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15 };
  return t[i];
}
// order of the g x g groups (g = 1, 2, 4, 8): returns gy * g + gx of the i-th group
__device__ __forceinline__ int pattern_group(int scan_idx, int g, int i)
{
  if (scan_idx == 1) return i;
  if (scan_idx == 2) return (i % g) * g + (i / g);
  // up-right diagonal over a g x g grid: walk anti-diagonals from bottom-left to top-right
  int c = 0;
  for (int d = 0; d < 2 * g - 1; ++d) {
    const int y0 = d < g ? d : g - 1;
    const int cnt = (d < g) ? d + 1 : 2 * g - 1 - d;
    if (i < c + cnt) { const int y = y0 - (i - c); return y * g + (d - y); }
    c += cnt;
  }
  return 0;
}
__device__ __forceinline__ int scan_pos(int scan_idx, int log2_size, int idx)
{
  const int n = 1 << log2_size;
  if (log2_size == 2) return pattern_pos4(scan_idx, idx);
  const int g = n >> 2;
  const int grp = pattern_group(scan_idx, g, idx >> 4);
  const int p = pattern_pos4(scan_idx, idx & 15);
  return ((grp / g) * 4 + (p >> 2)) * n + (grp % g) * 4 + (p & 3);
}

// One coefficient group (16 coefficients in scan order, positions pos16) of the sign-hiding pass, quant-generic.c:82-156.
// A group reads and changes only its own coefficients; what it needs from the rest of the block is whether it is the
// "last" group -- the highest one in scan order that holds a non-zero level (last_cg, :99-101, :153).
template <typename CP, typename QP>
__device__ __forceinline__ void sign_hide_cg(CP coef, QP q_coef, const int (&pos16)[16], bool is_last_cg, const quant_consts &k)
{
  const int q_bits8 = k.q_bits - 8;
  auto delta_u = [&](int pos) -> int {
    const int qc = k.qtable ? k.qtable[pos] : k.flat_qc;
    const int c = coef[pos];
    const long long prod = (long long)(c < 0 ? -c : c) * qc;
    const int level = (int)((prod + k.add) >> k.q_bits);
    return (int)((prod - (long long)(int)((unsigned)level << k.q_bits)) >> q_bits8);
  };
  int first_nz = 16, last_nz = -1, abssum = 0;
  for (int n = 15; n >= 0; --n) if (q_coef[pos16[n]]) { last_nz = n; break; }
  for (int n = 0; n < 16; ++n) if (q_coef[pos16[n]]) { first_nz = n; break; }
  for (int n = first_nz; n <= last_nz; ++n) abssum += q_coef[pos16[n]];
  if (last_nz - first_nz < 4) return;
  const int signbit = q_coef[pos16[first_nz]] > 0 ? 0 : 1;
  if (signbit == (abssum & 1)) return;
  int min_cost_inc = 0x7fffffff, min_pos = -1, cur_cost = 0x7fffffff;
  int final_change = 0, cur_change = 0;
  for (int n = (is_last_cg ? last_nz : 15); n >= 0; --n) {
    const int pos = pos16[n];
    const int q = q_coef[pos];
    if (q != 0) {
      const int du = delta_u(pos);
      if (du > 0) { cur_cost = -du; cur_change = 1; }
      else if (n == first_nz && (q == 1 || q == -1)) { cur_cost = 0x7fffffff; }
      else { cur_cost = du; cur_change = -1; }
    } else if (n < first_nz && ((coef[pos] >= 0) ? 0 : 1) != signbit) {
      cur_cost = 0x7fffffff;
    } else { cur_cost = -delta_u(pos); cur_change = 1; }
    if (cur_cost < min_cost_inc) { min_cost_inc = cur_cost; final_change = cur_change; min_pos = pos; }
  }
  const int qm = q_coef[min_pos];
  if (qm == 32767 || qm == -32768) final_change = -1;
  if (coef[min_pos] >= 0) q_coef[min_pos] = (i16)(qm + final_change);
  else q_coef[min_pos] = (i16)(qm - final_change);
}

// the whole block on one thread, groups from the last to the first like the reference (used inside the fused kernel)
template <typename CP, typename QP>
__device__ void sign_hide_block(CP coef, QP q_coef, int width, int scan_idx, const quant_consts &k)
{
  const int log2_size = width == 4 ? 2 : width == 8 ? 3 : width == 16 ? 4 : 5;
  const int n_coef = width * width;
  unsigned ac_sum = 0;
  for (int n = 0; n < n_coef; ++n) ac_sum += (unsigned)quant_level(coef[n], k.qtable ? k.qtable[n] : k.flat_qc, k);
  if (ac_sum < 2) return;
  bool seen_nz = false;
  for (int subset = (n_coef - 1) >> 4; subset >= 0; --subset) {
    int pos16[16];
    bool nz = false;
#pragma unroll
    for (int n = 0; n < 16; ++n) { pos16[n] = scan_pos(scan_idx, log2_size, (subset << 4) + n); nz = nz || q_coef[pos16[n]] != 0; }
    sign_hide_cg(coef, q_coef, pos16, nz && !seen_nz, k);
    seen_nz = seen_nz || nz;
  }
}

// Batched form: one LANE per coefficient group -- NCG = (width / 4)^2 adjacent lanes share a block.  The block-level facts a
// group needs come from its neighbours in the wave: ac_sum (:52-68) is a sum over the block's lanes, "last group" a
// ballot of the lanes that hold a non-zero level.  (The first version ran one thread per BLOCK.)
template <int NCG>
__global__ __launch_bounds__(256) void sign_hide_kernel(const i16 *__restrict__ coef, i16 *__restrict__ q_coef,
                                                        size_t count, int width, int scan_idx, quant_consts k)
{
  const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t blk = item / NCG;
  const int cg = (int)(item % NCG), lane = threadIdx.x & 63;
  const bool valid = blk < count;
  const size_t off = (valid ? blk : count - 1) * (size_t)(width * width);
  const i16 *c = coef + off;
  i16 *q = q_coef + off;
  const int log2_size = width == 4 ? 2 : width == 8 ? 3 : width == 16 ? 4 : 5;
  int pos16[16];
  bool nz = false;
  u32 ac = 0;
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    pos16[n] = scan_pos(scan_idx, log2_size, (cg << 4) + n);
    nz = nz || q[pos16[n]] != 0;
    ac += (u32)quant_level(c[pos16[n]], k.qtable ? k.qtable[pos16[n]] : k.flat_qc, k);
  }
  ac = group_sum<NCG>(ac);
  const unsigned long long bal = __ballot(nz && valid);
  // lanes of my block above me: bits lane + 1 .. base + NCG - 1
  const int base = lane & ~(NCG - 1);
  const unsigned long long group = (NCG == 64 ? ~0ull : ((1ull << NCG) - 1ull)) << base;
  const unsigned long long above = group & ~((2ull << lane) - 1ull);
  const bool is_last = nz && (bal & above) == 0ull;
  if (valid && ac >= 2) sign_hide_cg(c, q, pos16, is_last, k);
}

// coeff_abs_sum (quant-generic.c:323-330): one wave per block
__global__ __launch_bounds__(256) void coeff_abs_sum_kernel(const i16 *__restrict__ c, size_t length, size_t count, u32 *__restrict__ sums)
{
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  for (size_t b = wave; b < count; b += nwaves) {
    const i16 *p = c + b * length;
    u32 acc = 0;
    for (size_t i = lane; i < length; i += 64) { int v = p[i]; acc += (u32)(v < 0 ? -v : v); }
    acc = group_sum<64>(acc);
    if (lane == 0) sums[b] = acc;
  }
}

// ---------------------------------------------------------------------------
// Fused kvz_quantize_residual (rdoq off), quant-generic.c:180-273.  256/N TUs per
// workgroup, N threads per TU; everything between the loads of ref/pred and the
// stores of rec/coeff stays in LDS and registers.
// ---------------------------------------------------------------------------
template <int N, int TRK>     // TRK: 0 DCT, 2 DST, 4 transform skip
__global__ __launch_bounds__(256) void quantize_residual_kernel(const u8 *__restrict__ ref_in, const u8 *pred_in,
                                                                u8 *rec_out, i16 *__restrict__ coeff_out,
                                                                i32 *__restrict__ has_coeffs, size_t count,
                                                                int scan_order, quant_consts k,
                                                                u32 *__restrict__ ssd_out, u32 *__restrict__ abs_sum_out)
{
  constexpr int TPB = 256 / N;
  constexpr int LD = lds_tile_ld(N);                 // odd number of dwords per row: bank-conflict free (transform_core.h)
  constexpr int LOG2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  constexpr int TS_SHIFT = 15 - 8 - LOG2N;
  __shared__ __attribute__((aligned(16))) i16 sa[TPB * N * LD];     // residual / coefficients
  __shared__ __attribute__((aligned(16))) i16 sb[TPB * N * LD];     // transform scratch
  __shared__ __attribute__((aligned(16))) i16 sq[TPB * N * LD];     // quantized coefficients (same padded rows)
  __shared__ int s_has[TPB];

  __shared__ __attribute__((aligned(16))) u8 sp[TPB * N * N];       // prediction, overwritten in place by the reconstruction

  const int tid = threadIdx.x, tu = tid / N, row = tid % N;
  const size_t ngroups = (count + TPB - 1) / TPB;
  i16 *a = sa + tu * N * LD, *b = sb + tu * N * LD, *q = sq + tu * N * LD;
  u8 *pp = sp + tu * N * N + row * N;
  constexpr int PCH = TPB * N * N / 16;              // 16-pixel chunks per group (64 .. 512)

  for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const size_t first = g * TPB;
    const size_t blk = first + tu;
    const bool valid = blk < count;
    // stage in: coalesced 16-byte loads of ref and pred; residual (ref - pred) as int16 into `a`, pred kept in LDS
#pragma unroll
    for (int c = tid; c < PCH; c += 256) {
      const int e = c * 16, t = e / (N * N), w = e % (N * N);
      uint4 rv = make_uint4(0, 0, 0, 0), pv = rv;
      if (first + t < count) {
        rv = *(const uint4 *)(ref_in + first * (size_t)(N * N) + e);
        pv = *(const uint4 *)(pred_in + first * (size_t)(N * N) + e);
      }
      *(uint4 *)(sp + e) = pv;
      const u32 rr[4] = { rv.x, rv.y, rv.z, rv.w }, pq[4] = { pv.x, pv.y, pv.z, pv.w };
#pragma unroll
      for (int j = 0; j < 2; ++j) {                  // 8 residuals = one 16-byte LDS store
        union { uint4 v; i16 s[8]; } o;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          o.s[i] = (i16)((int)((rr[2 * j + (i >> 2)] >> (8 * (i & 3))) & 255u) - (int)((pq[2 * j + (i >> 2)] >> (8 * (i & 3))) & 255u));
        const int ww = w + 8 * j;
        lds_tile_store8<N, LD>(sa + t * N * LD, ww, o.v);
      }
    }
    if (row == 0) s_has[tu] = 0;
    __syncthreads();
    if constexpr (TRK == 4) {
#pragma unroll
      for (int x = 0; x < N; ++x) a[row * LD + x] = (i16)((int)a[row * LD + x] << TS_SHIFT);
    } else {
      transform_2d_lds<N, TRK, LD>(a, b, row);
    }
    __syncthreads();
    // quantize own row
    int any = 0;
#pragma unroll
    for (int x = 0; x < N; ++x) {
      const int n = row * N + x;
      const int v = quant_one(a[row * LD + x], k.qtable ? k.qtable[n] : k.flat_qc, k);
      q[row * LD + x] = (i16)v;
      any |= v;
    }
    if (k.signhide) {
      __syncthreads();
      if (row == 0) {
        struct lds_view { i16 *p; int ld, n; __device__ i16 &operator[](int i) const { return p[(i / n) * ld + (i % n)]; } };
        lds_view cv = { a, LD, N }, qv = { q, LD, N };
        sign_hide_block(cv, qv, N, scan_order, k);
      }
      __syncthreads();
      any = 0;
#pragma unroll
      for (int x = 0; x < N; ++x) any |= q[row * LD + x];
    }
    if (any) atomicOr(&s_has[tu], 1);
    __syncthreads();
    const int has = s_has[tu];
    // dequantize own row back into `a`, inverse transform (done for every TU so that
    // barriers stay uniform; the result is only used when has != 0)
#pragma unroll
    for (int x = 0; x < N; ++x) a[row * LD + x] = (i16)dequant_one(q[row * LD + x], row * N + x, k);
    __syncthreads();
    if constexpr (TRK == 4) {
#pragma unroll
      for (int x = 0; x < N; ++x) a[row * LD + x] = (i16)(((int)a[row * LD + x] + (1 << (TS_SHIFT - 1))) >> TS_SHIFT);
    } else {
      transform_2d_lds<N, TRK + 1, LD>(a, b, row);
    }
    __syncthreads();
    if (has) {                                       // reconstruction over the prediction, in place
#pragma unroll
      for (int x = 0; x < N; ++x) {
        const i16 val = (i16)((int)a[row * LD + x] + (int)pp[x]);
        pp[x] = (u8)(val < 0 ? 0 : (val > 255 ? 255 : val));
      }
    }
    if (valid && row == 0) has_coeffs[blk] = has;
    if (ssd_out) {
      // rd=0 TU cost inputs without re-reading anything: kvz_pixels_calc_ssd(ref, rec) (search.c:291) and
      // kvz_coeff_abs_sum (rdo.c:219); the N threads of a TU are consecutive lanes
      // the original row comes back from L2 (this workgroup streamed it a moment ago): keeping a copy in LDS
      // would cost the plain entry a workgroup of occupancy
      u32 rw[N / 4];
#pragma unroll
      for (int j = 0; j < N / 4; ++j) rw[j] = valid ? ((const u32 *)(ref_in + blk * (size_t)(N * N) + row * N))[j] : 0u;
      u32 sq2 = 0, sab = 0;
#pragma unroll
      for (int x = 0; x < N; ++x) {
        const int dd = (int)((rw[x >> 2] >> (8 * (x & 3))) & 255u) - (int)pp[x];
        const int qq = q[row * LD + x];
        sq2 += (u32)(dd * dd);
        sab += (u32)(qq < 0 ? -qq : qq);
      }
      sq2 = group_sum<N>(sq2);
      sab = group_sum<N>(sab);
      if (valid && row == 0) { ssd_out[blk] = sq2; abs_sum_out[blk] = sab; }
    }
    __syncthreads();
    // stage out: coalesced 16-byte stores of the reconstruction and of the quantized coefficients
#pragma unroll
    for (int c = tid; c < PCH; c += 256) {
      const int e = c * 16, t = e / (N * N);
      if (first + t < count) *(uint4 *)(rec_out + first * (size_t)(N * N) + e) = *(const uint4 *)(sp + e);
    }
#pragma unroll
    for (int c = tid; c < 2 * PCH; c += 256) {
      const int e = c * 8, t = e / (N * N);
      if (first + t < count) *(uint4 *)(coeff_out + first * (size_t)(N * N) + e) = lds_tile_load8<N, LD>(sq + t * N * LD, e % (N * N));
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// 4x4 TUs: one lane per TU, the whole kvz_quantize_residual chain (quant-generic.c:180-273, rdoq off, no sign
// hiding) in registers.  A TU is 16 bytes of ref and of pred: a lane's loads and stores are single 16-byte
// accesses and a wave's are fully coalesced, so there is no LDS and no barrier.  (The LDS kernel below needed
// nine barriers per 64 TUs and reached 2.2 TB/s.)
// ---------------------------------------------------------------------------
// FLAT: flat scaling (no per-coefficient table), compiled without the table loads -- with them in the loop every branch that
// may have issued one ends in a wait on vector memory, which would also wait for the prefetch.
template <int TRK, bool PIPE = true, bool FLAT = false>     // TRK: 0 DCT, 2 DST (intra luma), 4 transform skip
__global__ __launch_bounds__(256) void quantize_residual4_lane_kernel(const u8 *__restrict__ ref_in, const u8 *pred_in, u8 *rec_out,
                                                                      i16 *__restrict__ coeff_out, i32 *__restrict__ has_coeffs,
                                                                      size_t count, quant_consts k,
                                                                      u32 *__restrict__ ssd_out, u32 *__restrict__ abs_sum_out)
{
  constexpr int TS_SHIFT = 15 - 8 - 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // Software pipeline (wait_vmem_all, kvz_hip_internal.h): the next TU's 32 bytes are requested before this one is worked on, and
  // the iteration's one wait on vector memory sits just before its stores -- the plain loop waited at its top for the loads
  // it had just issued AND for the stores of the TU before.  PIPE false: that plain loop, for A/B runs ("pipe" 0).
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 rv = make_uint4(0, 0, 0, 0), pv = rv, rn = rv, pn = rv;
  constexpr bool PF = PIPE && FLAT;                       // the prefetch only where nothing else in the loop loads
  if (PF && i < count) { rv = ld_stream_u4(ref_in + i * 16); pv = *(const uint4 *)(pred_in + i * 16); }
  if (PF) wait_vmem_all();
  for (; i < count; i += stride) {
    if (PF) {
      const size_t in = i + stride;
      if (in < count) { rn = ld_stream_u4(ref_in + in * 16); pn = *(const uint4 *)(pred_in + in * 16); }
    } else {
      rv = ld_stream_u4(ref_in + i * 16);
      pv = *(const uint4 *)(pred_in + i * 16);
    }
    const u32 rr[4] = { rv.x, rv.y, rv.z, rv.w }, pq[4] = { pv.x, pv.y, pv.z, pv.w };
    int res[4][4], c[4][4];
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x) res[y][x] = (int)((rr[y] >> (8 * x)) & 255u) - (int)((pq[y] >> (8 * x)) & 255u);
    if (TRK == 4) {
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) c[y][x] = (int)(i16)(res[y][x] << TS_SHIFT);
    } else {
      // dct-generic.c:567-576: two passes, each row -> transposed column (transform_core.h pass_1d)
      int t1[4][4], w[4];
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        pass_1d<4, TRK>(res[y], w, 1);
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) t1[kx][y] = w[kx];
      }
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        pass_1d<4, TRK>(t1[y], w, 8);
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) c[kx][y] = w[kx];
      }
    }
    int q[4][4], any = 0;
    u32 sab = 0;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int n = y * 4 + x;
        q[y][x] = quant_one(c[y][x], (!FLAT && k.qtable) ? k.qtable[n] : k.flat_qc, k);
        any |= q[y][x];
        sab += (u32)(q[y][x] < 0 ? -q[y][x] : q[y][x]);
      }
    const bool has = any != 0;
    u32 out[4] = { pq[0], pq[1], pq[2], pq[3] };
    if (has) {
      int d[4][4], r2[4][4];
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x)
          d[y][x] = FLAT ? (int)(i16)clip16((int)((unsigned)__mul24(q[y][x], k.dq_scale) + (unsigned)k.dq_add) >> k.dq_shift)
                         : (int)(i16)dequant_one(q[y][x], y * 4 + x, k);
      if (TRK == 4) {
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
          for (int x = 0; x < 4; ++x) r2[y][x] = (int)(i16)((d[y][x] + (1 << (TS_SHIFT - 1))) >> TS_SHIFT);
      } else {
        // dct-generic.c:578-587: pass 1 columns of the input -> rows of tmp, pass 2 columns of tmp -> rows of out
        int tmp[4][4], v[4], w[4];
#pragma unroll
        for (int col = 0; col < 4; ++col) {
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) v[kx] = d[kx][col];
          pass_1d<4, TRK + 1>(v, w, 7);
#pragma unroll
          for (int x = 0; x < 4; ++x) tmp[col][x] = w[x];
        }
#pragma unroll
        for (int col = 0; col < 4; ++col) {
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) v[kx] = tmp[kx][col];
          pass_1d<4, TRK + 1>(v, w, 12);
#pragma unroll
          for (int x = 0; x < 4; ++x) r2[col][x] = w[x];
        }
      }
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        u32 wv = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          int val = (int)(i16)(r2[y][x] + (int)((pq[y] >> (8 * x)) & 255u));             // quant-generic.c:255
          // Keep the clamp away from the shift that produced r2: hipcc 7.2 fuses "(a >> 16) clamped to a byte, two at a
          // time" into v_ashr_pk_u8_i32 and then ORs the next byte into the result assuming its upper half is zero,
          // but gfx950 leaves the destination's upper 16 bits untouched -- every third pixel of a row came out
          // OR-ed with the first (tests/test_abi.py keeps the instruction out of the build).
          asm volatile("" : "+v"(val));
          wv |= (u32)(val < 0 ? 0 : (val > 255 ? 255 : val)) << (8 * x);
        }
        out[y] = wv;
      }
    }
    if (PF) { wait_vmem_all(); rv = rn; pv = pn; }          // rr / pq hold this TU's pixels: the registers can take the next one's
    *(uint4 *)(rec_out + i * 16) = make_uint4(out[0], out[1], out[2], out[3]);
    uint4 c0, c1;
    c0.x = (u32)(q[0][0] & 0xffff) | ((u32)q[0][1] << 16); c0.y = (u32)(q[0][2] & 0xffff) | ((u32)q[0][3] << 16);
    c0.z = (u32)(q[1][0] & 0xffff) | ((u32)q[1][1] << 16); c0.w = (u32)(q[1][2] & 0xffff) | ((u32)q[1][3] << 16);
    c1.x = (u32)(q[2][0] & 0xffff) | ((u32)q[2][1] << 16); c1.y = (u32)(q[2][2] & 0xffff) | ((u32)q[2][3] << 16);
    c1.z = (u32)(q[3][0] & 0xffff) | ((u32)q[3][1] << 16); c1.w = (u32)(q[3][2] & 0xffff) | ((u32)q[3][3] << 16);
    st_stream_u4(coeff_out + i * 16, c0);
    st_stream_u4(coeff_out + i * 16 + 8, c1);
    has_coeffs[i] = has ? 1 : 0;
    if (ssd_out) {
      u32 sq2 = 0;
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int dd = (int)((rr[y] >> (8 * x)) & 255u) - (int)((out[y] >> (8 * x)) & 255u);
          sq2 += (u32)(dd * dd);
        }
      ssd_out[i] = sq2;
      abs_sum_out[i] = sab;
    }
  }
}

// quant-generic.c:196-204 / :253-259 as stand-alone elementwise kernels (the RDOQ route of the drop-in)
__global__ __launch_bounds__(256) void residual_kernel(const u8 *__restrict__ ref_in, const u8 *__restrict__ pred_in, i16 *__restrict__ res, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) res[i] = (i16)((int)ref_in[i] - (int)pred_in[i]);
}
__global__ __launch_bounds__(256) void reconstruct_kernel(const i16 *__restrict__ res, const u8 *pred_in, u8 *rec_out, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const i16 val = (i16)((int)res[i] + (int)pred_in[i]);
    rec_out[i] = (u8)(val < 0 ? 0 : (val > 255 ? 255 : val));
  }
}

namespace kvzhip {
int launch_quantize_residual_tile(int n, const u8 *ref_in, const u8 *pred_in, u8 *rec_out, i16 *coeff_out, i32 *has_coeffs, size_t count,
                                  int q_bits, int add, int flat_qc, const int32_t *qtable, int dq_mode, int dq_shift, int dq_add,
                                  int dq_scale, const int32_t *dqtable, u32 *ssd_out, u32 *abs_sum_out, hipStream_t st);
int launch_quantize_residual8_reg(const u8 *ref_in, const u8 *pred_in, u8 *rec_out, i16 *coeff_out, i32 *has_coeffs, size_t count,
                                  int q_bits, int add, int flat_qc, int dq_shift, int dq_add, int dq_scale,
                                  u32 *ssd_out, u32 *abs_sum_out, hipStream_t st);
}

extern "C" {

int kvz_hip_quant_batch(const kvz_hip_quant_params *p, const kvz_hip_coeff *coef, kvz_hip_coeff *q_coef,
                        int width, int type, int scan_idx, size_t count, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  quant_consts k;
  if (!p || !coef || !q_coef || scan_idx < 0 || scan_idx > 2 || !make_consts(p, width, type, type, &k)) return kvzhip::invalid_arg(__func__);
  if ((((uintptr_t)coef | (uintptr_t)q_coef) & 15) != 0) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  hipStream_t st = ctx_stream(s);
  const size_t total = count * (size_t)(width * width);
  hipLaunchKernelGGL(quant_kernel, dim3(stream_grid(total, 2048, (unsigned)tuning("quant_wgs_per_cu", 256))), dim3(256), 0, st, coef, q_coef, total, width * width, k);
  KVZ_CHECK_LAUNCH("quant_kernel");
  if (k.signhide) {
    const size_t ncg = (size_t)(width / 4) * (width / 4), items = count * ncg;
    const dim3 grid((unsigned)((items + 255) / 256));
    if (items > 0xffffffffull * 256ull) return kvzhip::invalid_arg(__func__);
    switch (width) {
      case 4: hipLaunchKernelGGL(sign_hide_kernel<1>, grid, dim3(256), 0, st, coef, q_coef, count, width, scan_idx, k); break;
      case 8: hipLaunchKernelGGL(sign_hide_kernel<4>, grid, dim3(256), 0, st, coef, q_coef, count, width, scan_idx, k); break;
      case 16: hipLaunchKernelGGL(sign_hide_kernel<16>, grid, dim3(256), 0, st, coef, q_coef, count, width, scan_idx, k); break;
      default: hipLaunchKernelGGL(sign_hide_kernel<64>, grid, dim3(256), 0, st, coef, q_coef, count, width, scan_idx, k); break;
    }
    KVZ_CHECK_LAUNCH("sign_hide_kernel");
  }
  return KVZ_HIP_OK;
}

int kvz_hip_dequant_batch(const kvz_hip_quant_params *p, const kvz_hip_coeff *q_coef, kvz_hip_coeff *coef,
                          int width, int type, size_t count, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  quant_consts k;
  if (!p || !coef || !q_coef || !make_consts(p, width, type, type, &k)) return kvzhip::invalid_arg(__func__);
  if ((((uintptr_t)coef | (uintptr_t)q_coef) & 15) != 0) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  const size_t total = count * (size_t)(width * width);
  hipLaunchKernelGGL(dequant_kernel, dim3(stream_grid(total, 2048, (unsigned)tuning("quant_wgs_per_cu", 256))), dim3(256), 0, ctx_stream(s), q_coef, coef, total, width * width, k);
  KVZ_CHECK_LAUNCH("dequant_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_coeff_abs_sum_batch(const kvz_hip_coeff *coeffs, size_t length, size_t count, uint32_t *sums, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!coeffs || !sums) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  hipLaunchKernelGGL(coeff_abs_sum_kernel, dim3(stream_grid(count, 4)), dim3(256), 0, ctx_stream(s), coeffs, length, count, sums);
  KVZ_CHECK_LAUNCH("coeff_abs_sum_kernel");
  return KVZ_HIP_OK;
}

static int quantize_residual_impl(const kvz_hip_quant_params *p, int cu_is_intra, int width, int color, int scan_order,
                                  int use_trskip, const kvz_hip_pixel *ref_in, const kvz_hip_pixel *pred_in,
                                  kvz_hip_pixel *rec_out, kvz_hip_coeff *coeff_out, int32_t *has_coeffs,
                                  uint32_t *ssd_out, uint32_t *abs_sum_out, size_t count, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  quant_consts k;
  // quant uses type 0 / 2, dequant 0 / 2 / 3 (quant-generic.c:224, :244)
  const int tq = color == 0 ? 0 : 2, tdq = color == 0 ? 0 : (color == 1 ? 2 : 3);
  if (!p || !ref_in || !pred_in || !rec_out || !coeff_out || !has_coeffs || color < 0 || color > 2 ||
      scan_order < 0 || scan_order > 2 || !make_consts(p, width, tq, tdq, &k)) return kvzhip::invalid_arg("kvz_hip_quantize_residual_batch / kvz_hip_quantize_residual_cost_batch");
  if ((((uintptr_t)ref_in | (uintptr_t)pred_in | (uintptr_t)rec_out | (uintptr_t)coeff_out) & 15) != 0) return kvzhip::invalid_arg("kvz_hip_quantize_residual_batch / kvz_hip_quantize_residual_cost_batch");
  if (count == 0) return KVZ_HIP_OK;
  hipStream_t st = ctx_stream(s);
  const bool dst = (width == 4 && color == 0 && cu_is_intra);     // strategies-dct.c:66-85
  if ((width == 32 || width == 16) && !use_trskip && !k.signhide && tuning("qr_tile_kernel", 1))
    return launch_quantize_residual_tile(width, ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k.q_bits, k.add, k.flat_qc, k.qtable,
                                         k.dq_mode, k.dq_shift, k.dq_add, k.dq_scale, k.dqtable, ssd_out, abs_sum_out, st);
  // 8x8: sixteen TUs per matrix-core tile (quant_tile_mfma.hip); the cost variant and "qr8_tile_kernel" = 0 use the register kernel
  if (width == 8 && !use_trskip && !k.signhide && !ssd_out && tuning("qr_tile_kernel", 1) && tuning("qr8_tile_kernel", 1))
    return launch_quantize_residual_tile(8, ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k.q_bits, k.add, k.flat_qc, k.qtable,
                                         k.dq_mode, k.dq_shift, k.dq_add, k.dq_scale, k.dqtable, nullptr, nullptr, st);
  if (width == 8 && !use_trskip && !k.signhide && !k.qtable && k.dq_mode == 0 && tuning("qr8_reg_kernel", 1))
    return launch_quantize_residual8_reg(ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k.q_bits, k.add, k.flat_qc,
                                         k.dq_shift, k.dq_add, k.dq_scale, ssd_out, abs_sum_out, st);
  if (width == 4 && !k.signhide && tuning("qr4_lane_kernel", 1)) {
    const unsigned grid = stream_grid(count, 256, (unsigned)tuning("qr4_wgs_per_cu", 96)       /* measured: 16: 4.8 TB/s, 64: 5.4, 128: 5.4 */);
    const bool flat = !k.qtable && k.dq_mode == 0;
    // measured A/B on one box (0.5 GiB operands): with the in-wave prefetch 5.00-5.05 TB/s, without 5.15-5.21: at 8 waves per SIMD the
    // hardware's wave interleaving already hides the latency and the prefetch costs a wave of occupancy (72 against 60 VGPRs)
    const bool pipe = tuning("pipe", 0) != 0;
#define KVZ_QR4(T) do { if (flat && pipe) hipLaunchKernelGGL((quantize_residual4_lane_kernel<T, true, true>), dim3(grid), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k, ssd_out, abs_sum_out); \
                        else if (flat) hipLaunchKernelGGL((quantize_residual4_lane_kernel<T, false, true>), dim3(grid), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k, ssd_out, abs_sum_out); \
                        else hipLaunchKernelGGL((quantize_residual4_lane_kernel<T, false, false>), dim3(grid), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k, ssd_out, abs_sum_out); } while (0)
    if (use_trskip) KVZ_QR4(4); else if (dst) KVZ_QR4(2); else KVZ_QR4(0);
#undef KVZ_QR4
    KVZ_CHECK_LAUNCH("quantize_residual4_lane_kernel");
    return KVZ_HIP_OK;
  }
#define KVZ_QR(N, TRK) hipLaunchKernelGGL((quantize_residual_kernel<N, TRK>), dim3(stream_grid(count, 256 / N, (unsigned)tuning("qr_wgs_per_cu", N == 8 ? 64 : 16))), dim3(256), 0, st, \
                                          ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, scan_order, k, ssd_out, abs_sum_out)
  switch (width) {
    case 4: if (use_trskip) KVZ_QR(4, 4); else if (dst) KVZ_QR(4, 2); else KVZ_QR(4, 0); break;
    case 8: if (use_trskip) KVZ_QR(8, 4); else KVZ_QR(8, 0); break;
    case 16: if (use_trskip) KVZ_QR(16, 4); else KVZ_QR(16, 0); break;
    case 32: if (use_trskip) KVZ_QR(32, 4); else KVZ_QR(32, 0); break;
    default: return kvzhip::invalid_arg("kvz_hip_quantize_residual_batch / kvz_hip_quantize_residual_cost_batch");
  }
#undef KVZ_QR
  KVZ_CHECK_LAUNCH("quantize_residual_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_residual_batch(const kvz_hip_pixel *ref_in, const kvz_hip_pixel *pred_in, kvz_hip_coeff *residual, size_t n, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (n == 0) return KVZ_HIP_OK;
  if (!ref_in || !pred_in || !residual) return kvzhip::invalid_arg(__func__);
  hipLaunchKernelGGL(residual_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, ctx_stream(s), ref_in, pred_in, residual, n);
  KVZ_CHECK_LAUNCH("residual_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_reconstruct_batch(const kvz_hip_coeff *residual, const kvz_hip_pixel *pred_in, kvz_hip_pixel *rec_out, size_t n, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (n == 0) return KVZ_HIP_OK;
  if (!residual || !pred_in || !rec_out) return kvzhip::invalid_arg(__func__);
  hipLaunchKernelGGL(reconstruct_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, ctx_stream(s), residual, pred_in, rec_out, n);
  KVZ_CHECK_LAUNCH("reconstruct_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_quantize_residual_batch(const kvz_hip_quant_params *p, int cu_is_intra, int width, int color, int scan_order,
                                    int use_trskip, const kvz_hip_pixel *ref_in, const kvz_hip_pixel *pred_in,
                                    kvz_hip_pixel *rec_out, kvz_hip_coeff *coeff_out, int32_t *has_coeffs,
                                    size_t count, kvz_hip_stream s)
{
  return quantize_residual_impl(p, cu_is_intra, width, color, scan_order, use_trskip, ref_in, pred_in, rec_out, coeff_out, has_coeffs,
                                nullptr, nullptr, count, s);
}

int kvz_hip_quantize_residual_cost_batch(const kvz_hip_quant_params *p, int cu_is_intra, int width, int color, int scan_order,
                                         int use_trskip, const kvz_hip_pixel *ref_in, const kvz_hip_pixel *pred_in,
                                         kvz_hip_pixel *rec_out, kvz_hip_coeff *coeff_out, int32_t *has_coeffs,
                                         uint32_t *ssd_out, uint32_t *coeff_abs_sum_out, size_t count, kvz_hip_stream s)
{
  if (!ssd_out || !coeff_abs_sum_out) { set_error_msg("kvz_hip_quantize_residual_cost_batch: null cost buffer"); return KVZ_HIP_ERR_INVALID; }
  return quantize_residual_impl(p, cu_is_intra, width, color, scan_order, use_trskip, ref_in, pred_in, rec_out, coeff_out, has_coeffs,
                                ssd_out, coeff_abs_sum_out, count, s);
}

}  // extern "C"
