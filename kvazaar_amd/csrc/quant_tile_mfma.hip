// quant_tile_mfma.hip -- fused kvz_quantize_residual for 32x32, 16x16 and 8x8 TUs on the matrix cores: one 32x32 pixel
// "tile" per wave step = one 32x32 TU, FOUR 16x16 TUs arranged 2 x 2, or SIXTEEN 8x8 TUs arranged 4 x 4, with the same
// instruction stream (8x8: ~25 vector instructions per TU where the register kernel of quant8_reg.hip spends 35 and is
// bound by their issue).
//
// Reference: src/strategies/generic/quant-generic.c:180-273 (rdoq off, no transform skip, sign hiding off -- the other
// variants stay on quantize_residual_kernel in quant.hip) around the transform pairs of dct-generic.c:368-597
// (partial_butterfly_16 / _32 and inverses; shifts log2N - 1, log2N + 6 and 7, 12).
//
// Method (dct32_mfma.hip): a 1-D pass is an exact integer matrix product; the int16 operand is split into byte
// planes X = 256 Xh + Xl' + 128 and each pass is two v_mfma_i32_32x32x32_i8 (+ a constant).  For 16x16 the 32x32
// coefficient matrix is diag(M16, M16): the tile's four quadrants transform independently, so every lane and every
// accumulator register of the MFMA tile is live (the previous 16x16 kernel put two TUs in a tile with half of K
// dead and ran at 2.95 TB/s).
//
// The fused kernels are bound by VECTOR-INSTRUCTION ISSUE, not by HBM, LDS or the matrix pipe (rocprofv3 PMC,
// profiles/r02_a_qr_pmc_before.txt: one vector instruction per ~4 cycles per SIMD; matrix pipe 15 % busy), so the
// pipeline is arranged to need no data movement between the passes and as few vector instructions as possible:
//   * operand roles alternate so that each pass's accumulator tile (16 rows in registers, column = lane) is the next
//     pass's operand as it stands:  S rows (lane = pixel row) --A--> T'[j][k] --B--> C[x][k] (lane = coefficient COLUMN)
//     -> quant -> dequant --A--> U[k][j'] --B--> residual[i'][j'] (lane = pixel row again, the layout of the prediction).
//     The previous kernels kept coefficients row-per-lane and transposed through the matrix core before the inverse
//     (2 MFMA + 40 vector instructions per tile);
//   * plane-offset and rounding constants enter through the MFMA's C operand, loaded from LDS tables: the recombination
//     is (hi << 8) + lo, then the shift -- two vector instructions per value per pass;
//   * quantisation is signed (quant8_reg.hip): 4 instructions per coefficient; clip16 + packing are one
//     v_cvt_pk_i16_i32 per pair; reconstruction is packed int16;
//   * the quantised coefficients leave in column layout through the wave's LDS tile: ds_write_b16 per value (LDS issue
//     slots are free), read back as 16-byte row chunks, stored coalesced.
// HBM traffic per tile: 1 KiB ref + 1 KiB pred + 1 KiB rec + 2 KiB coeff = 5*N*N per TU.
#include "dct32_mfma_core.h"

using namespace kvzhip;

namespace {

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

// the tile's 32x32 coefficient matrix: M32, or diag(M16, M16)
template <int N>
__host__ __device__ constexpr int mx(int a, int b)
{
  return N == 32 ? dct_coef(32, a, b) : ((a / N == b / N) ? dct_coef(N, a % N, b % N) : 0);
}
__host__ __device__ constexpr int kap(int h, int e) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// Constant MFMA operands of every lane and the C-operand constants of the four passes, built at compile time.
//   t_kap: bytes mx(r, kappa(h, e))      forward pass 1 (B) and pass 2 (A)
//   t_col: bytes mx(kappa(h, e), r)      inverse pass 1 (B) and pass 2 (A)
//   c[p][lane][g]: plane offset 128 * (row / column sum of mx) + rounding of pass p for accumulator register g
template <int N>
struct tile_table {
  u32 t_kap[64][4], t_col[64][4];
  int c[4][64][16];
  constexpr tile_table() : t_kap(), t_col(), c()
  {
    constexpr int LOG2N = N == 32 ? 5 : (N == 16 ? 4 : 3);
    int rowsum[32] = {}, colsum[32] = {};
    for (int a = 0; a < 32; ++a)
      for (int b = 0; b < 32; ++b) { rowsum[a] += mx<N>(a, b); colsum[b] += mx<N>(a, b); }
    for (int lane = 0; lane < 64; ++lane) {
      const int r = lane & 31, h = lane >> 5;
      for (int q = 0; q < 4; ++q) {
        u32 a = 0, b = 0;
        for (int i = 0; i < 4; ++i) {
          const int e = 4 * q + i;
          a |= ((u32)mx<N>(r, kap(h, e)) & 255u) << (8 * i);
          b |= ((u32)mx<N>(kap(h, e), r) & 255u) << (8 * i);
        }
        t_kap[lane][q] = a; t_col[lane][q] = b;
      }
      for (int g = 0; g < 16; ++g) {
        c[0][lane][g] = 128 * rowsum[r] + (1 << (LOG2N - 2));            // forward pass 1: column k = lane, shift log2N - 1
        c[1][lane][g] = 128 * rowsum[kap(h, g)] + (1 << (LOG2N + 5));    // forward pass 2: row x = register, shift log2N + 6
        c[2][lane][g] = 128 * colsum[r] + (1 << 6);                      // inverse pass 1: column j' = lane, shift 7
        c[3][lane][g] = 128 * colsum[kap(h, g)] + (1 << 11);             // inverse pass 2: row i' = register, shift 12
      }
    }
  }
};
static __constant__ tile_table<32> c_tile32 = tile_table<32>();
static __constant__ tile_table<16> c_tile16 = tile_table<16>();
static __constant__ tile_table<8> c_tile8 = tile_table<8>();
template <int N> __device__ __forceinline__ const tile_table<N> &tile_tab();
template <> __device__ __forceinline__ const tile_table<32> &tile_tab<32>() { return c_tile32; }
template <> __device__ __forceinline__ const tile_table<16> &tile_tab<16>() { return c_tile16; }
template <> __device__ __forceinline__ const tile_table<8> &tile_tab<8>() { return c_tile8; }

struct qt_consts {
  int q_bits, add, flat_qc;
  const int32_t *qtable;
  int dq_mode, dq_shift, dq_add, dq_scale;
  const int32_t *dqtable;
};

// natural (lane (r,h): columns 16h .. 16h+15 as 4 dwords) <-> kappa order (dword q = columns 8q+4h .. +3).
__device__ __forceinline__ void kappa_swap(u32 (&a)[4])
{
  const u32x2v p = __builtin_amdgcn_permlane32_swap(a[0], a[1], false, false);
  const u32x2v q = __builtin_amdgcn_permlane32_swap(a[2], a[3], false, false);
  a[0] = p.x; a[2] = p.y; a[1] = q.x; a[3] = q.y;
}
__device__ __forceinline__ void kappa_unswap(u32 (&a)[4])
{
  const u32x2v p = __builtin_amdgcn_permlane32_swap(a[0], a[2], false, false);
  const u32x2v q = __builtin_amdgcn_permlane32_swap(a[1], a[3], false, false);
  a[0] = p.x; a[1] = p.y; a[2] = q.x; a[3] = q.y;
}

// the C operand of a pass: four 16-byte LDS reads (every table entry is read by exactly one lane: conflict-free)
__device__ __forceinline__ i32x16 c_init(const u32x4v *tab, int lane)
{
  const u32x4v a = tab[lane], b = tab[64 + lane], c = tab[128 + lane], d = tab[192 + lane];
  const i32x16 v = { (int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w,
                     (int)c.x, (int)c.y, (int)c.z, (int)c.w, (int)d.x, (int)d.y, (int)d.z, (int)d.w };
  return v;
}

// 16 accumulator values (hi, lo products; lo already carries the constants) -> value >> shift, two instructions each
__device__ __forceinline__ void combine(const i32x16 &hi, const i32x16 &lo, int shift, int (&o)[16])
{
#pragma unroll
  for (int g = 0; g < 16; ++g) o[g] = ((hi[g] << 8) + lo[g]) >> shift;
}
// byte planes of 16 int32 taken modulo 2^16 (the forward passes' (short) cast)
__device__ __forceinline__ void planes_wrap(const int (&t)[16], op16 &hi, op16 &lo) { planes_from_regs(t, hi, lo, 0x80808080u); }
// byte planes of 16 int32 clipped to int16 (the inverse passes / dequantisation): v_cvt_pk_i16_i32 per pair, then the planes
__device__ __forceinline__ void planes_sat(const int (&t)[16], op16 &hi, op16 &lo)
{
  u32 d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pk_i16(t[2 * i], t[2 * i + 1]));
  planes_from_rows(d, hi, lo);
}

template <int N, bool COST, bool PIPE = true>
__global__ __launch_bounds__(256, 4) void quantize_residual_tile_kernel(const u8 *__restrict__ ref_in, const u8 *pred_in, u8 *rec_out,
                                                                     i16 *__restrict__ coeff_out, i32 *__restrict__ has_coeffs,
                                                                     size_t count, qt_consts k,
                                                                     u32 *__restrict__ ssd_out, u32 *__restrict__ abs_sum_out)
{
  constexpr int LOG2N = N == 32 ? 5 : (N == 16 ? 4 : 3);
  constexpr int TUS = (32 / N) * (32 / N);             // TUs per tile
  static_assert(!(N == 8 && COST), "the 8x8 cost variant stays on quant8_reg.hip");
  constexpr size_t TU_PX = (size_t)N * N;
  const tile_table<N> &tb = tile_tab<N>();
  __shared__ __attribute__((aligned(16))) u8 s_tile[4][2048];
  __shared__ __attribute__((aligned(16))) u32x4v s_c[4][256];          // [pass][quarter * 64 + lane] = 4 consecutive registers
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)wv;
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t ntiles = (count + TUS - 1) / TUS;
  u8 *tile = s_tile[wv];
  for (int i = threadIdx.x; i < 4 * 256; i += 256) {
    const int p = i >> 8, q = (i >> 6) & 3, l = i & 63;
    const u32x4v v = { (u32)tb.c[p][l][4 * q], (u32)tb.c[p][l][4 * q + 1], (u32)tb.c[p][l][4 * q + 2], (u32)tb.c[p][l][4 * q + 3] };
    s_c[p][q * 64 + l] = v;
  }
  op16 t_kap, t_col;
#pragma unroll
  for (int q = 0; q < 4; ++q) { t_kap.w[q] = tb.t_kap[lane][q]; t_col.w[q] = tb.t_col[lane][q]; }
  __syncthreads();
  const i32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  const bool flat = k.qtable == nullptr && k.dq_mode == 0;        // wave-uniform
  const int bias_pos = k.add, bias_neg = (int)((1u << k.q_bits) - 1u) - k.add;

  // this lane's 16-byte chunk of the tile's pixels: row r, columns 16h .. 16h+15.  16x16: TU (r >> 4) + 2h of the four, row r & 15
  // 8x8: the 16 pixels are row r & 7 of the two TUs (r >> 3) * 4 + 2h and + 1 of the tile's sixteen: two 8-byte accesses
  // (`live` bit 0 / bit 1 = the first / second of them exists)
  const int my_tu = N == 32 ? 0 : (N == 16 ? (r >> 4) + 2 * h : (r >> 3) * 4 + 2 * h);
  const size_t px_off = N == 32 ? (size_t)(2 * r + h) * 16 : (N == 16 ? (size_t)my_tu * 256 + (size_t)(r & 15) * 16 : (size_t)my_tu * 64 + (size_t)(r & 7) * 8);
  auto load = [&](size_t t, u32x4v &rv, u32x4v &pv, int &live) {
    const size_t tu = t * TUS + (size_t)my_tu;
    if (N == 8) {
      live = (tu < count ? 1 : 0) | (tu + 1 < count ? 2 : 0);
      const size_t last = (count - 1) * TU_PX + (size_t)(r & 7) * 8;          // a missing TU mirrors the last one
      const size_t a = (live & 1) ? t * TUS * TU_PX + px_off : last, b = (live & 2) ? t * TUS * TU_PX + px_off + 64 : last;
      const u32x2v ra = __builtin_nontemporal_load((const u32x2v *)(ref_in + a)), rb = __builtin_nontemporal_load((const u32x2v *)(ref_in + b));
      const u32x2v pa = *(const u32x2v *)(pred_in + a), pb = *(const u32x2v *)(pred_in + b);
      rv = u32x4v{ ra.x, ra.y, rb.x, rb.y };
      pv = u32x4v{ pa.x, pa.y, pb.x, pb.y };
      return;
    }
    live = tu < count ? 3 : 0;
    const size_t base = live ? t * TUS * TU_PX + px_off : (count - 1) * TU_PX + (px_off & 255);   // a missing TU mirrors the last one
    rv = __builtin_nontemporal_load((const u32x4v *)(ref_in + base));
    pv = *(const u32x4v *)(pred_in + base);
  };

  // Software pipeline (wait_vmem_all, kvz_hip_internal.h).  Left to the compiler, this loop waited for its OWN prefetch right
  // after issuing it (s_waitcnt vmcnt(0) at the loop top: the rotation of the prefetch registers was scheduled there) and
  // every iteration paid a full memory latency -- the 43 % of wave-cycles parked on memory of
  // profiles/r02_z_qr_pmc_after.txt.  The one vmcnt(0) of an iteration now sits just before the iteration's first store:
  // the prefetch was issued half an iteration of arithmetic earlier, the previous iteration's stores a whole one.
  size_t t = wave;
  u32x4v rv, pv, rn, pn;
  int live = 0, live_n = 0;
  if (t < ntiles) load(t, rv, pv, live);
  if (PIPE) wait_vmem_all();
  for (; t < ntiles; t += nwaves) {
    const size_t tn = t + nwaves;
    if (tn < ntiles) load(tn, rn, pn, live_n);          // prefetch the wave's next tile (never one being written)
    u32 rf[4] = { rv.x, rv.y, rv.z, rv.w }, pr[4] = { pv.x, pv.y, pv.z, pv.w };
    kappa_swap(rf);
    kappa_swap(pr);
    // residual (quant-generic.c:196-204), element e = 4q + i <-> column kappa(h, e); packed int16 pairs
    u32 d[8], p16[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      p16[2 * q] = __builtin_amdgcn_perm(0u, pr[q], 0x0c010c00u);
      p16[2 * q + 1] = __builtin_amdgcn_perm(0u, pr[q], 0x0c030c02u);
      d[2 * q] = __builtin_bit_cast(u32, __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, rf[q], 0x0c010c00u)) - __builtin_bit_cast(v2s, p16[2 * q]));
      d[2 * q + 1] = __builtin_bit_cast(u32, __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, rf[q], 0x0c030c02u)) - __builtin_bit_cast(v2s, p16[2 * q + 1]));
    }
    op16 hi, lo;
    planes_from_rows(d, hi, lo);
    int v[16];
    // forward pass 1: T'[j][k] = sum_n S[j][n] mx[k][n]; rows j in registers, column k = lane
    {
      const i32x16 ah = mfma_i8(hi, t_kap, zero), al = mfma_i8(lo, t_kap, c_init(s_c[0], lane));
      combine(ah, al, LOG2N - 1, v);
    }
    planes_wrap(v, hi, lo);
    // forward pass 2: C[x][k] = sum_j mx[x][j] T'[j][k]; rows x (vertical frequency) in registers, column k = lane
    {
      const i32x16 ah = mfma_i8(t_kap, hi, zero), al = mfma_i8(t_kap, lo, c_init(s_c[1], lane));
      combine(ah, al, LOG2N + 6, v);
    }
    // quant (quant-generic.c:55-67).  |coefficient| <= 32640 for pixel differences, so the reference's (short) cast is the
    // identity; flat scaling: signed form, |level| < 2^14, the clip to int16 never acts
    int lv[16];
    if (flat) {
#pragma unroll
      for (int g = 0; g < 16; ++g) lv[g] = (__mul24(v[g], k.flat_qc) + (v[g] < 0 ? bias_neg : bias_pos)) >> k.q_bits;
    } else {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int x = kap(h, g);
        const int n = N == 32 ? x * 32 + r : (x & (N - 1)) * N + (r & (N - 1));
        const int c = (int)(short)v[g], a = c < 0 ? -c : c;
        const int qc = k.qtable ? k.qtable[n] : k.flat_qc;
        const int level = (int)(((long long)a * qc + k.add) >> k.q_bits);
        lv[g] = clip16(c < 0 ? -level : level);
      }
    }
    // has_coeffs per TU: register group (row half of the coefficient) x lane group (column half)
    int any_top = 0, any_bot = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) { any_top |= lv[g]; any_bot |= lv[8 + g]; }
    // the iteration's one wait on vector memory (see above), then the prefetched registers become the next tile's
    if (PIPE) wait_vmem_all();          // PIPE false (tuning "qr_tile_pipe" 0): the compiler's own placement, for A/B runs
    const u32x4v rv_next = rn, pv_next = pn;
    const int live_next = live_n;
    bool has_row0, has_row1;                              // of this lane's PIXEL row half: column halves 0 and 1
    u32 has_q = 0;                                        // 8x8: bit q = the TU of this lane's pixel row block and column block q has coefficients
    unsigned long long any_tile;
    if (N == 32) {
      any_tile = __ballot((any_top | any_bot) != 0);
      has_row0 = has_row1 = any_tile != 0ull;
    } else if (N == 8) {
      // coefficient layout: TU (row block, column block) = (register quartet g >> 2, lane r >> 3); one ballot per quartet
      unsigned long long bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = __ballot((lv[4 * q] | lv[4 * q + 1] | lv[4 * q + 2] | lv[4 * q + 3]) != 0);
      any_tile = bq[0] | bq[1] | bq[2] | bq[3];
      has_row0 = has_row1 = false;
      // the lanes of column block c: r >> 3 == c in both halves
      const unsigned long long m0 = 0x000000ff000000ffull;
      {
        // pixel layout: this lane's row block r >> 3, column blocks q = 0..3 (its four output dwords)
        const int rb = r >> 3;
        const unsigned long long mine = rb == 0 ? bq[0] : rb == 1 ? bq[1] : rb == 2 ? bq[2] : bq[3];
#pragma unroll
        for (int q = 0; q < 4; ++q) has_q |= ((mine & (m0 << (8 * q))) != 0ull ? 1u : 0u) << q;
      }
      if (lane < 16) {
        const size_t tu = t * 16 + (size_t)lane;
        const int q = lane >> 2, cb = lane & 3;
        const unsigned long long b = q == 0 ? bq[0] : q == 1 ? bq[1] : q == 2 ? bq[2] : bq[3];
        if (tu < count) has_coeffs[tu] = (b & (m0 << (8 * cb))) != 0ull ? 1 : 0;
      }
    } else {
      const unsigned long long lo_lanes = 0x0000ffff0000ffffull;
      const unsigned long long bt = __ballot(any_top != 0), bb = __ballot(any_bot != 0);
      const bool ha = (bt & lo_lanes) != 0ull, hb = (bb & lo_lanes) != 0ull, hc = (bt & ~lo_lanes) != 0ull, hd = (bb & ~lo_lanes) != 0ull;
      any_tile = bt | bb;
      has_row0 = r < 16 ? ha : hb;                       // TU (pixel row half, column half 0)
      has_row1 = r < 16 ? hc : hd;                       // TU (pixel row half, column half 1)
      if (lane < 4) {
        const size_t tu = t * 4 + (size_t)lane;
        const bool f = lane == 0 ? ha : lane == 1 ? hb : lane == 2 ? hc : hd;
        if (tu < count) has_coeffs[tu] = f ? 1 : 0;
      }
    }
    if (N == 32 && lane == 0) has_coeffs[t] = any_tile != 0ull ? 1 : 0;

    // coefficients out: column layout -> the wave's LDS tile (row-major, swizzled 16-byte slots) -> coalesced row chunks
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int x = kap(h, g);
      *(i16 *)(tile + slot_of(4 * x + (r >> 3)) * 16 + (r & 7) * 2) = (i16)lv[g];
    }
    wave_lds_fence();
    if (N == 8) {
      // lane l: row l & 7 of TU (l >> 3) of the tile's first eight, and of the TU eight further: consecutive lanes store
      // consecutive 16-byte rows (2 x 1 KiB per wave, fully coalesced)
      const int tin = lane >> 3, row = lane & 7, x0 = (tin >> 2) * 8 + row, c = tin & 3;
      const u32x4v a = *(const u32x4v *)(tile + slot_of(4 * x0 + c) * 16);
      const u32x4v b = *(const u32x4v *)(tile + slot_of(4 * (x0 + 16) + c) * 16);
      wave_lds_fence();
      const size_t tu_a = t * 16 + (size_t)tin, tu_b = tu_a + 8;
      if (tu_a < count) __builtin_nontemporal_store(a, (u32x4v *)(coeff_out + tu_a * 64) + row);
      if (tu_b < count) __builtin_nontemporal_store(b, (u32x4v *)(coeff_out + tu_b * 64) + row);
    } else {
      const u32x4v a = *(const u32x4v *)(tile + slot_of(lane) * 16);
      const u32x4v b = *(const u32x4v *)(tile + slot_of(64 + lane) * 16);
      wave_lds_fence();
      if (N == 32) {
        __builtin_nontemporal_store(a, (u32x4v *)(coeff_out + t * 1024) + lane);
        __builtin_nontemporal_store(b, (u32x4v *)(coeff_out + t * 1024) + 64 + lane);
      } else {
        // chunk `lane`: row x = lane >> 2 (< 16), quarter lane & 3 -> TU 2 * (quarter >> 1) (+ 1 for rows 16 .. 31), 16-byte half quarter & 1
        const int x = lane >> 2, qt = lane & 3;
        const size_t tu_a = t * 4 + (size_t)(2 * (qt >> 1)), tu_b = tu_a + 1;
        const size_t off = (size_t)x * 32 + (size_t)(qt & 1) * 16;
        if (tu_a < count) __builtin_nontemporal_store(a, (u32x4v *)((u8 *)(coeff_out + tu_a * 256) + off));
        if (tu_b < count) __builtin_nontemporal_store(b, (u32x4v *)((u8 *)(coeff_out + tu_b * 256) + off));
      }
    }

    u32 out[4] = { pr[0], pr[1], pr[2], pr[3] };          // a TU without coefficients keeps its prediction (:262-271)
    if (any_tile != 0ull) {                               // wave-uniform
      // dequant (quant-generic.c:290-320) in the column layout; the clip to int16 happens in planes_sat
      int dv[16];
      if (flat) {
#pragma unroll
        for (int g = 0; g < 16; ++g) dv[g] = (__mul24(lv[g], k.dq_scale) + k.dq_add) >> k.dq_shift;
      } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int x = kap(h, g);
          const int n = N == 32 ? x * 32 + r : (x & (N - 1)) * N + (r & (N - 1));
          if (k.dq_mode == 0) dv[g] = (__mul24(lv[g], k.dq_scale) + k.dq_add) >> k.dq_shift;
          else if (k.dq_mode == 1) dv[g] = (lv[g] * k.dqtable[n] + k.dq_add) >> k.dq_shift;
          else dv[g] = (int)((u32)clip16(lv[g] * k.dqtable[n]) << k.dq_shift);
        }
      }
      planes_sat(dv, hi, lo);
      // inverse pass 1 (down the columns): U[k][j'] = sum_x C[x][k] mx[x][j']; rows k in registers, column j' = lane
      {
        const i32x16 ah = mfma_i8(hi, t_col, zero), al = mfma_i8(lo, t_col, c_init(s_c[2], lane));
        combine(ah, al, 7, v);
      }
      planes_sat(v, hi, lo);
      // inverse pass 2: residual[i'][j'] = sum_k mx[k][i'] U[k][j']; rows i' (pixel column) in registers, lane = pixel row j'
      {
        const i32x16 ah = mfma_i8(t_col, hi, zero), al = mfma_i8(t_col, lo, c_init(s_c[3], lane));
        combine(ah, al, 12, v);
      }
      // reconstruction: (int16)(clip16(residual) + pred) clipped to a pixel (quant-generic.c:253-259), packed
      u32 o16[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const v2s s = __builtin_amdgcn_cvt_pk_i16(v[2 * i], v[2 * i + 1]) + __builtin_bit_cast(v2s, p16[i]);
        const v2s z = { 0, 0 }, m = { 255, 255 };
        o16[i] = __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_elementwise_max(s, z), m));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool has = N == 8 ? ((has_q >> q) & 1u) != 0u : (q < 2 ? has_row0 : has_row1);
        if (has) out[q] = __builtin_amdgcn_perm(o16[2 * q + 1], o16[2 * q], 0x06040200u);
      }
    }
    if (COST) {
      // rd=0 TU cost inputs from the registers (search.c:291, rdo.c:219): SSD(ref, rec) in the pixel layout, sum |coeff| in
      // the coefficient layout; 16x16: register group = column half, lane row half (r >> 4) = the other half
      u32 ssd[2] = { 0, 0 }, sab[2] = { 0, 0 };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const u32 sel = half ? 0x0c030c02u : 0x0c010c00u;
          const v2s dd = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, rf[q], sel)) - __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, out[q], sel));
          ssd[q >> 1] = (u32)__builtin_amdgcn_sdot2(dd, dd, (int)ssd[q >> 1], false);
        }
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) sab[g >> 3] += (u32)(lv[g] < 0 ? -lv[g] : lv[g]);
      if (N == 32) {
        const u32 s = group_sum<64>(ssd[0] + ssd[1]), a = group_sum<64>(sab[0] + sab[1]);
        if (lane == 0) { ssd_out[t] = s; abs_sum_out[t] = a; }
      } else {
        // pixel layout: TU (row half r >> 4, column half = register group); both lane halves h hold columns of both halves
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          u32 s = group_sum<16>(ssd[ch]);
          s += (u32)__shfl_xor((int)s, 32, 64);
          const size_t tu = t * 4 + (size_t)((r >> 4) + 2 * ch);
          if ((lane == 0 || lane == 16) && tu < count) ssd_out[tu] = s;
          // coefficient layout: TU (row half = register group ch, column half = lane r >> 4)
          u32 a = group_sum<16>(sab[ch]);
          a += (u32)__shfl_xor((int)a, 32, 64);
          const size_t tuc = t * 4 + (size_t)(ch + 2 * (r >> 4));
          if ((lane == 0 || lane == 16) && tuc < count) abs_sum_out[tuc] = a;
        }
      }
    }
    kappa_unswap(out);
    if (N == 8) {
      const size_t base = t * TUS * TU_PX + px_off;
      if (live & 1) *(u32x2v *)(rec_out + base) = u32x2v{ out[0], out[1] };
      if (live & 2) *(u32x2v *)(rec_out + base + 64) = u32x2v{ out[2], out[3] };
    } else if (live) {
      const u32x4v ov = { out[0], out[1], out[2], out[3] };
      const size_t base = t * TUS * TU_PX + px_off;
      *(u32x4v *)(rec_out + base) = ov;
    }
    rv = rv_next; pv = pv_next; live = live_next;
  }
}

template <int N>
int launch_tile(const u8 *ref_in, const u8 *pred_in, u8 *rec_out, i16 *coeff_out, i32 *has_coeffs, size_t count, const qt_consts &k,
                u32 *ssd_out, u32 *abs_sum_out, hipStream_t st)
{
  constexpr size_t TUS = (size_t)(32 / N) * (32 / N);
  const size_t ntiles = (count + TUS - 1) / TUS;
  size_t wgs = (ntiles + 3) / 4;
  // workgroups per CU, measured at 0.5 GiB operands: 32x32 -- 8: 5.47, 16: 5.12, 32: 4.88, 64: 4.49 TB/s; 16x16 -- 8: 5.32, 16: 5.13, 32: 4.90, 64: 4.47
  const size_t cap = (size_t)num_cus() * (size_t)tuning(N == 32 ? "qr32_wgs_per_cu" : (N == 16 ? "qr16_wgs_per_cu" : "qr8_wgs_per_cu"), 8);
  if (wgs > cap) wgs = cap;
  if constexpr (N == 8) {
    hipLaunchKernelGGL((quantize_residual_tile_kernel<8, false>), dim3((unsigned)wgs), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out,
                       has_coeffs, count, k, ssd_out, abs_sum_out);
  } else if (!ssd_out && !tuning("qr_tile_pipe", 1)) {
    hipLaunchKernelGGL((quantize_residual_tile_kernel<N, false, false>), dim3((unsigned)wgs), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out,
                       has_coeffs, count, k, ssd_out, abs_sum_out);
  } else if (ssd_out)
    hipLaunchKernelGGL((quantize_residual_tile_kernel<N, true>), dim3((unsigned)wgs), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out,
                       has_coeffs, count, k, ssd_out, abs_sum_out);
  else
    hipLaunchKernelGGL((quantize_residual_tile_kernel<N, false>), dim3((unsigned)wgs), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out,
                       has_coeffs, count, k, ssd_out, abs_sum_out);
  KVZ_CHECK_LAUNCH("quantize_residual_tile_kernel");
  return KVZ_HIP_OK;
}

}  // namespace

namespace kvzhip {
// consts are produced by quant.hip's make_consts (same field meaning)
int launch_quantize_residual_tile(int n, const u8 *ref_in, const u8 *pred_in, u8 *rec_out, i16 *coeff_out, i32 *has_coeffs, size_t count,
                                  int q_bits, int add, int flat_qc, const int32_t *qtable, int dq_mode, int dq_shift, int dq_add,
                                  int dq_scale, const int32_t *dqtable, u32 *ssd_out, u32 *abs_sum_out, hipStream_t st)
{
  const qt_consts k = { q_bits, add, flat_qc, qtable, dq_mode, dq_shift, dq_add, dq_scale, dqtable };
  if (n == 32) return launch_tile<32>(ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k, ssd_out, abs_sum_out, st);
  if (n == 8) return launch_tile<8>(ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k, nullptr, nullptr, st);
  return launch_tile<16>(ref_in, pred_in, rec_out, coeff_out, has_coeffs, count, k, ssd_out, abs_sum_out, st);
}
}  // namespace kvzhip
