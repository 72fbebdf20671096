// transform_core.h -- HEVC integer DCT / DST cores as exact int32 arithmetic in
// registers.  Shared by dct.hip and quant.hip (fused quantize_residual).
//
// Reference: src/strategies/generic/dct-generic.c.  The coefficient matrix
// (dct-generic.c:34-108) is generated at compile time from the first column of
// the 32-point matrix and the cosine symmetries; the even/odd recursion below is
// the same exact factorisation the reference's partial_butterfly_N_generic
// (:243-511) uses, so every int32 intermediate is identical.
#pragma once

#include "kvz_hip_internal.h"

namespace kvzhip {

// first column of the 32-point HEVC core transform
__host__ __device__ constexpr int dct_c32(int m)
{
  constexpr int c[32] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                          64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4 };
  return c[m];
}

// M_n[k][i], n in {2,4,8,16,32}
__host__ __device__ constexpr int dct_coef(int n, int k, int i)
{
  int m = ((k * (32 / n)) * (2 * i + 1)) % 128;
  int sign = 1;
  if (m > 64) m = 128 - m;
  if (m > 32) { m = 64 - m; sign = -1; }
  return m == 32 ? 0 : sign * dct_c32(m);
}

// HEVC 4x4 DST-VII (dct-generic.c:26-32)
__host__ __device__ constexpr int dst_coef(int k, int i)
{
  constexpr int d[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };
  return d[k * 4 + i];
}

#if defined(__HIPCC__)

// 24-bit multiply-add: every product here is (|coef| <= 90) x (<= 21-bit value)
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

// y[k] = sum_i M_N[k][i] * x[i]
template <int N>
struct fwd_core {
  static __device__ __forceinline__ void run(const int (&x)[N], int (&y)[N])
  {
    int e[N / 2], o[N / 2], ye[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { e[i] = x[i] + x[N - 1 - i]; o[i] = x[i] - x[N - 1 - i]; }
    fwd_core<N / 2>::run(e, ye);
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
      y[2 * k] = ye[k];
      int acc = 0;
#pragma unroll
      for (int i = 0; i < N / 2; ++i) acc = mad24(dct_coef(N, 2 * k + 1, i), o[i], acc);
      y[2 * k + 1] = acc;
    }
  }
};
template <>
struct fwd_core<2> {
  static __device__ __forceinline__ void run(const int (&x)[2], int (&y)[2])
  {
    y[0] = 64 * (x[0] + x[1]);
    y[1] = 64 * (x[0] - x[1]);
  }
};

// x[i] = sum_k M_N[k][i] * y[k]
template <int N>
struct inv_core {
  static __device__ __forceinline__ void run(const int (&y)[N], int (&x)[N])
  {
    int ye[N / 2], E[N / 2];
#pragma unroll
    for (int k = 0; k < N / 2; ++k) ye[k] = y[2 * k];
    inv_core<N / 2>::run(ye, E);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      int O = 0;
#pragma unroll
      for (int k = 0; k < N / 2; ++k) O = mad24(dct_coef(N, 2 * k + 1, i), y[2 * k + 1], O);
      x[i] = E[i] + O;
      x[N - 1 - i] = E[i] - O;
    }
  }
};
template <>
struct inv_core<2> {
  static __device__ __forceinline__ void run(const int (&y)[2], int (&x)[2])
  {
    x[0] = 64 * (y[0] + y[1]);
    x[1] = 64 * (y[0] - y[1]);
  }
};

// 4-point DST (dct-generic.c:206-240): forward y[k] = sum_i D[k][i] x[i],
// inverse x[i] = sum_k D[k][i] y[k]
__device__ __forceinline__ void dst4_fwd(const int (&x)[4], int (&y)[4])
{
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = mad24(dst_coef(k, i), x[i], acc);
    y[k] = acc;
  }
}
__device__ __forceinline__ void dst4_inv(const int (&y)[4], int (&x)[4])
{
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = mad24(dst_coef(k, i), y[k], acc);
    x[i] = acc;
  }
}

__device__ __forceinline__ int clip16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// One 1-D pass over a vector held in registers.
//   KIND 0 DCT / 2 DST forward: out = (short)((y + add) >> shift)   (truncating cast)
//   KIND 1 IDCT / 3 IDST:       out = clip16((x + add) >> shift)
template <int N, int KIND>
__device__ __forceinline__ void pass_1d(const int (&in)[N], int (&out)[N], int shift)
{
  const int add = 1 << (shift - 1);
  int t[N];
  if (KIND == 0) fwd_core<N>::run(in, t);
  else if (KIND == 1) inv_core<N>::run(in, t);
  else if (KIND == 2) { if (N == 4) dst4_fwd((const int (&)[4])in, (int (&)[4])t); }
  else { if (N == 4) dst4_inv((const int (&)[4])in, (int (&)[4])t); }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int v = (t[i] + add) >> shift;
    out[i] = (KIND & 1) ? clip16(v) : (int)(short)v;
  }
}

// Row stride (int16) of an N x N tile in LDS: N + 2 makes a row an ODD number of dwords, so the
// row-per-thread accesses of transform_2d_lds (lane = row, consecutive lanes N + 2 int16 apart,
// consecutive tiles N rows apart) fall on distinct banks.  With the power-of-two strides used at
// first, rocprofv3 showed SQ_LDS_BANK_CONFLICT at 75 % of the LDS cycles of quantize_residual 8x8.
constexpr int lds_tile_ld(int n) { return n + 2; }

// 8 consecutive elements (e = multiple of 8) of a tile as four dwords; pairs never straddle a row
template <int N, int LD>
__device__ __forceinline__ void lds_tile_store8(i16 *tile, int e, uint4 v)
{
  const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = e + 2 * j;
    *(u32 *)(tile + (idx / N) * LD + (idx % N)) = w[j];
  }
}
template <int N, int LD>
__device__ __forceinline__ uint4 lds_tile_load8(const i16 *tile, int e)
{
  u32 w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = e + 2 * j;
    w[j] = *(const u32 *)(tile + (idx / N) * LD + (idx % N));
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// 2-D transform of one N x N block by N cooperating threads (thread `row` owns
// one row/column).  `a` holds the input block in LDS (row stride LD), `b` is an
// LDS scratch of the same shape; the result is left in `a` (row-major).  The
// caller brackets the call with barriers for loading/storing `a`.
//   forward (dct-generic.c:567-576): pass 1 rows of the input -> columns of tmp,
//   pass 2 rows of tmp -> columns of out; shifts log2N-1, log2N+6.
//   inverse (:578-587): pass 1 columns of the input -> rows of tmp, pass 2
//   columns of tmp -> rows of out; shifts 7, 12.
template <int N, int KIND, int LD>
__device__ __forceinline__ void transform_2d_lds(i16 *a, i16 *b, int row)
{
  constexpr int LOG2N = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
  int v[N], w[N];
  if ((KIND & 1) == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = a[row * LD + i];
    pass_1d<N, KIND>(v, w, LOG2N - 1);
#pragma unroll
    for (int k = 0; k < N; ++k) b[k * LD + row] = (i16)w[k];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = b[row * LD + i];
    pass_1d<N, KIND>(v, w, LOG2N + 6);
#pragma unroll
    for (int k = 0; k < N; ++k) a[k * LD + row] = (i16)w[k];
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = a[k * LD + row];
    pass_1d<N, KIND>(v, w, 7);
#pragma unroll
    for (int i = 0; i < N; ++i) b[row * LD + i] = (i16)w[i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = b[k * LD + row];
    pass_1d<N, KIND>(v, w, 12);
    // every thread has finished reading its column of `a` before pass 1 wrote b,
    // and the barrier above orders those reads before these writes
#pragma unroll
    for (int i = 0; i < N; ++i) a[row * LD + i] = (i16)w[i];
  }
}

#endif  // __HIPCC__
}  // namespace kvzhip
