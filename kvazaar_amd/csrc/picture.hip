// picture.hip -- SAD / SATD / SSD / bipred-blend kernels for gfx950.
//
// Reference semantics: src/strategies/generic/picture-generic.c and the macros
// of src/strategies/strategies-picture.h (cited per kernel).  Everything here is
// integer/byte streaming work bounded by HBM bandwidth: wide coalesced loads,
// v_sad_u8 / packed-int16 butterflies in registers, DPP reductions, one
// coalesced store per wave.  No LDS is needed for the contiguous-block kernels
// (there is no reuse to exploit: every byte is read exactly once).
#include "kvz_hip_internal.h"
#include "satd_regs.h"

using namespace kvzhip;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Streaming (read-once) loads: the nontemporal policy keeps once-read bytes from
// displacing useful lines; measured +4..10 % HBM read rate on MI355X (tools/bw_probe.hip).
__device__ __forceinline__ uint4 ld_stream16(const u8 *p)
{
  u32x4 v = __builtin_nontemporal_load((const u32x4 *)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 ld_stream8(const u8 *p)
{
  u32x2 v = __builtin_nontemporal_load((const u32x2 *)p);
  return make_uint2(v.x, v.y);
}

__device__ __forceinline__ u32 sad_dword(u32 a, u32 b, u32 acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
__device__ __forceinline__ u32 sad16(uint4 a, uint4 b)
{
  u32 s = sad_dword(a.x, b.x, 0);
  s = sad_dword(a.y, b.y, s);
  s = sad_dword(a.z, b.z, s);
  return sad_dword(a.w, b.w, s);
}

// ---------------------------------------------------------------------------
// sad_NxN over contiguous block pairs (picture-generic.c:460-486) and the dual
// variant (:497-519).  A "chunk" is 16 bytes; a block has L = N*N/16 chunks.
// Each wave streams 64*U consecutive chunks per iteration (1 KiB per load
// instruction, U loads per array in flight), reduces the L lanes of a block
// with DPP and writes the wave's results with one coalesced store.
// ---------------------------------------------------------------------------
template <int N, int U, bool DUAL>
__global__ __launch_bounds__(256) void sad_nxn_kernel(const u8 *__restrict__ a, const u8 *__restrict__ b,
                                                      u32 *__restrict__ costs, size_t count,
                                                      size_t pred_stride, size_t item_stride)
{
  constexpr int L = N * N / 16;                 // chunks (lanes) per block: 1, 4, 16, 64, 256
  constexpr int LW = L > 64 ? 64 : L;           // lanes of one wave that share a block
  constexpr int UB = L > 64 ? L / 64 : 1;       // wave-loads that make up one block (N = 64: 4)
  static_assert(U % UB == 0, "U must cover whole blocks");
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t total_chunks = count * L;        // count = number of (block1, block2) pairs
  constexpr size_t CH = (size_t)64 * U;

  for (size_t base = wave * CH; base < total_chunks; base += nwaves * CH) {
    u32 s[U];
    // wave-uniform: a full tile issues all 2*U loads back to back (no per-load
    // bounds branch, so nothing forces an early s_waitcnt); only the last,
    // ragged tile takes the guarded path.
    const bool full = base + CH <= total_chunks;
    uint4 x[U], y[U];
    if (full && !DUAL) {
      const u8 *pa = a + (base + lane) * 16, *pb = b + (base + lane) * 16;
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = ld_stream16(pa + u * 1024);
#pragma unroll
      for (int u = 0; u < U; ++u) y[u] = ld_stream16(pb + u * 1024);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t c = base + (size_t)u * 64 + lane;
        const size_t cc = c < total_chunks ? c : total_chunks - 1;
        if (DUAL) {
          const size_t blk = cc / L, within = cc % L;      // blk = 2*item + k
          x[u] = ld_stream16(a + (blk >> 1) * item_stride + (blk & 1) * pred_stride + within * 16);
          y[u] = ld_stream16(b + (blk >> 1) * (size_t)(N * N) + within * 16);
        } else {
          x[u] = ld_stream16(a + cc * 16);
          y[u] = ld_stream16(b + cc * 16);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t c = base + (size_t)u * 64 + lane;
      s[u] = (full || c < total_chunks) ? sad16(x[u], y[u]) : 0u;
    }
    if (L == 1) {                                // N == 4: every chunk is a whole block
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t c = base + (size_t)u * 64 + lane;
        if (c < count) costs[c] = s[u];
      }
      continue;
    }
    // fold the UB loads of one block, then the LW lanes
    constexpr int R = U / UB;                    // results per lane-group per iteration
    u32 r[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      u32 t = 0;
#pragma unroll
      for (int j = 0; j < UB; ++j) t += s[i * UB + j];
      r[i] = group_sum<LW>(t);
    }
    // lane (g = lane / LW, p = lane % LW) stores result p of group g
    const int g = lane / LW, p = lane % LW;
    if (p < R) {
      u32 v = r[0];
#pragma unroll
      for (int i = 1; i < R; ++i) v = (p == i) ? r[i] : v;
      // block index of (iteration-result p, group g)
      const size_t blk = (base + (size_t)p * UB * 64) / L + g;
      if (blk < count) costs[blk] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// SATD.  One lane owns one 8x8 sub-block (or one 4x4 block): 128 input bytes,
// Hadamard in packed int16 registers (|coef| <= 64*255 fits), no cross-lane
// traffic until the final per-block sum.  The last butterfly stage is folded
// into the absolute sum: |a+b| + |a-b| = 2*max(|a|,|b|).
// picture-generic.c:240-328 (8x8: (sum+2)>>2), :105-196 (4x4: (sum+1)>>1).
// ---------------------------------------------------------------------------
// strategies-picture.h:40-56 (SATD_NxN) / picture-generic.c:357-390 (dual).
// Lane = one 8x8 sub-block; the (N/8)^2 lanes of a block are consecutive.
template <int N, bool DUAL>
__global__ __launch_bounds__(256) void satd_nxn_kernel(const u8 *__restrict__ a, const u8 *__restrict__ b,
                                                       u32 *__restrict__ costs, size_t count,
                                                       size_t pred_stride, size_t item_stride)
{
  constexpr int W8 = N / 8, SB = W8 * W8;       // 1, 4, 16, 64
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const size_t total = count * SB;
  const size_t total_up = (total + 63) & ~(size_t)63;     // keep whole waves in the loop (DPP)
  for (size_t i = tid; i < total_up; i += nthreads) {
    const bool valid = i < total;
    const size_t ii = valid ? i : total - 1;              // tail lanes re-read the last sub-block (result discarded)
    const size_t blk = ii / SB;
    const int sidx = (int)(ii % SB), sy = sidx / W8, sx = sidx % W8;
    const u8 *pa, *pb;
    if (DUAL) {
      pa = a + (blk >> 1) * item_stride + (blk & 1) * pred_stride;
      pb = b + (blk >> 1) * (size_t)(N * N);
    } else {
      pa = a + blk * (size_t)(N * N);
      pb = b + blk * (size_t)(N * N);
    }
    pa += (size_t)(sy * 8) * N + sx * 8;
    pb += (size_t)(sy * 8) * N + sx * 8;
    u32 ra[16], rb[16];
    if (N == 8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint4 t = *(const uint4 *)(pa + 16 * k), w = *(const uint4 *)(pb + 16 * k);
        ra[4 * k] = t.x; ra[4 * k + 1] = t.y; ra[4 * k + 2] = t.z; ra[4 * k + 3] = t.w;
        rb[4 * k] = w.x; rb[4 * k + 1] = w.y; rb[4 * k + 2] = w.z; rb[4 * k + 3] = w.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        uint2 t = *(const uint2 *)(pa + (size_t)r * N), w = *(const uint2 *)(pb + (size_t)r * N);
        ra[2 * r] = t.x; ra[2 * r + 1] = t.y; rb[2 * r] = w.x; rb[2 * r + 1] = w.y;
      }
    }
    u32 v = satd8x8_regs(ra, rb);
    if (!valid) v = 0;
    v = group_sum<SB>(v);
    if (valid && sidx == 0) costs[blk] = v;
  }
}

// ---------------------------------------------------------------------------
// satd_8x8 over contiguous block pairs, the headline kernel: same fully coalesced
// streaming as sad_nxn_kernel (lane = one 16-byte chunk = rows 2p, 2p+1 of a block,
// 4 lanes per block, U chunks per array in flight per lane).  The Hadamard runs
// on packed int16: horizontal stages and the row-pair stage in registers, the
// two remaining vertical stages across the quad with DPP quad_perm moves.  A lane
// whose stage bit is set computes (partner - own) instead of (own - partner):
// the sign of a whole coefficient never matters under the absolute sum.
// ---------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void satd8_kernel(const u8 *__restrict__ a, const u8 *__restrict__ b,
                                                    u32 *__restrict__ costs, size_t count)
{
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t total_chunks = count * 4;
  constexpr size_t CH = (size_t)64 * U;
  const short sg1 = (lane & 1) ? (short)-1 : (short)1, sg2 = (lane & 2) ? (short)-1 : (short)1;
  const v2s m1 = { sg1, sg1 }, m2 = { sg2, sg2 };

  for (size_t base = wave * CH; base < total_chunks; base += nwaves * CH) {
    const bool full = base + CH <= total_chunks;      // wave-uniform
    uint4 x[U], y[U];
    if (full) {
      const u8 *pa = a + (base + lane) * 16, *pb = b + (base + lane) * 16;
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = ld_stream16(pa + u * 1024);
#pragma unroll
      for (int u = 0; u < U; ++u) y[u] = ld_stream16(pb + u * 1024);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t c = base + (size_t)u * 64 + lane;
        const size_t cc = c < total_chunks ? c : total_chunks - 1;   // whole quads are in or out: count*4 chunks
        x[u] = ld_stream16(a + cc * 16);
        y[u] = ld_stream16(b + cc * 16);
      }
    }
    u32 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      u32 m = satd8_quad_part(x[u], y[u], m1, m2);
      m += dpp_mov<0xB1>(m);
      m += dpp_mov<0x4E>(m);
      r[u] = (m + 2) >> 2;
    }
    // lane (g = lane >> 2, p = lane & 3) stores result p of quad g: one coalesced 256-byte store per wave
    const int g = lane >> 2, p = lane & 3;
    if (p < U) {
      u32 v = r[0];
#pragma unroll
      for (int i = 1; i < U; ++i) v = (p == i) ? r[i] : v;
      const size_t blk = (base + (size_t)p * 64) / 4 + g;
      if (blk < count) costs[blk] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// satd_16x16 with the same streaming scheme: lane = one 16-byte chunk = one row
// of a 16x16 block (16 lanes per block) = one row of its left and of its right
// 8x8 sub-block.  Horizontal stages in registers; the three vertical stages cross
// 8 lanes: quad_perm for row bits 0/1, a row_shl:4 / row_shr:4 pair (bank-masked)
// for row bit 2.  Per-sub-block rounding (sum+2)>>2 is applied before the four
// sub-blocks are added (strategies-picture.h:40-56).
// ---------------------------------------------------------------------------
// x, y: row `lane & 15` of the block in each array.  Returns the block's SATD in every lane of its 16-lane row.
__device__ __forceinline__ u32 satd16_row_part(uint4 x, uint4 y, v2s m1, v2s m2, v2s m4)
{
  v2s d[8];
  d[0] = unpack_lo(x.x) - unpack_lo(y.x); d[1] = unpack_hi(x.x) - unpack_hi(y.x);
  d[2] = unpack_lo(x.y) - unpack_lo(y.y); d[3] = unpack_hi(x.y) - unpack_hi(y.y);
  d[4] = unpack_lo(x.z) - unpack_lo(y.z); d[5] = unpack_hi(x.z) - unpack_hi(y.z);
  d[6] = unpack_lo(x.w) - unpack_lo(y.w); d[7] = unpack_hi(x.w) - unpack_hi(y.w);
  u32 m[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {          // left / right 8x8 sub-block
    v2s *e = d + 4 * s;
    v2s s0 = e[0] + e[2], s1 = e[1] + e[3], f0 = e[0] - e[2], f1 = e[1] - e[3];     // column bits 2, 1
    v2s w[4] = { s0 + s1, s0 - s1, f0 + f1, f0 - f1 };
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v2s t = dpp_v2s<0xB1>(w[i]);       // row bit 0
      v2s u = w[i] * m1 + t;
      t = dpp_v2s<0x4E>(u);              // row bit 1
      u = u * m2 + t;
      t = dpp_xor4_v2s(u);               // row bit 2
      u = u * m4 + t;
      acc = abs_last_stage(u, acc);      // column bit 0 and the absolute sum
    }
    // sum over the 8 rows of the sub-block, then its rounding
    acc += dpp_mov<0xB1>(acc);
    acc += dpp_mov<0x4E>(acc);
    acc += dpp_xor4_u32(acc);
    m[s] = (acc + 2) >> 2;
  }
  u32 r = m[0] + m[1];                   // top (lanes 0-7) or bottom (lanes 8-15) pair of sub-blocks
  r += dpp_mov<0x128>(r);                // row_ror:8 -> the other pair
  return r;
}

template <int U>
__global__ __launch_bounds__(256) void satd16_kernel(const u8 *__restrict__ a, const u8 *__restrict__ b,
                                                     u32 *__restrict__ costs, size_t count)
{
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t total_chunks = count * 16;
  constexpr size_t CH = (size_t)64 * U;
  const short sg1 = (lane & 1) ? (short)-1 : (short)1, sg2 = (lane & 2) ? (short)-1 : (short)1, sg4 = (lane & 4) ? (short)-1 : (short)1;
  const v2s m1 = { sg1, sg1 }, m2 = { sg2, sg2 }, m4 = { sg4, sg4 };

  for (size_t base = wave * CH; base < total_chunks; base += nwaves * CH) {
    const bool full = base + CH <= total_chunks;
    uint4 x[U], y[U];
    if (full) {
      const u8 *pa = a + (base + lane) * 16, *pb = b + (base + lane) * 16;
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = ld_stream16(pa + u * 1024);
#pragma unroll
      for (int u = 0; u < U; ++u) y[u] = ld_stream16(pb + u * 1024);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t c = base + (size_t)u * 64 + lane;
        const size_t cc = c < total_chunks ? c : total_chunks - 1;
        x[u] = ld_stream16(a + cc * 16);
        y[u] = ld_stream16(b + cc * 16);
      }
    }
    u32 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = satd16_row_part(x[u], y[u], m1, m2, m4);
    const int g = lane >> 4, p = lane & 15;          // 4 blocks per wave-load
    if (p < U) {
      u32 v = r[0];
#pragma unroll
      for (int i = 1; i < U; ++i) v = (p == i) ? r[i] : v;
      const size_t blk = (base + (size_t)p * 64) / 16 + g;
      if (blk < count) costs[blk] = v;
    }
  }
}

// satd_4x4 streaming: lane = one block pair per load, U loads per array in flight
template <int U>
__global__ __launch_bounds__(256) void satd4_kernel(const u8 *__restrict__ a, const u8 *__restrict__ b,
                                                    u32 *__restrict__ costs, size_t count)
{
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  constexpr size_t CH = (size_t)64 * U;
  for (size_t base = wave * CH; base < count; base += nwaves * CH) {
    const bool full = base + CH <= count;
    uint4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t c = base + (size_t)u * 64 + lane;
      const size_t cc = (full || c < count) ? c : count - 1;
      x[u] = ld_stream16(a + cc * 16);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t c = base + (size_t)u * 64 + lane;
      const size_t cc = (full || c < count) ? c : count - 1;
      y[u] = ld_stream16(b + cc * 16);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t c = base + (size_t)u * 64 + lane;
      u32 ra[4] = { x[u].x, x[u].y, x[u].z, x[u].w }, rb[4] = { y[u].x, y[u].y, y[u].z, y[u].w };
      const u32 v = satd4x4_regs(ra, rb);
      if (full || c < count) costs[c] = v;
    }
  }
}

// satd_4x4 / satd_4x4_dual: lane = one 16-byte block pair
template <bool DUAL>
__global__ __launch_bounds__(256) void satd_4x4_kernel(const u8 *__restrict__ a, const u8 *__restrict__ b,
                                                       u32 *__restrict__ costs, size_t count,
                                                       size_t pred_stride, size_t item_stride)
{
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (size_t i = tid; i < count; i += nthreads) {
    uint4 x, y;
    if (DUAL) {
      x = *(const uint4 *)(a + (i >> 1) * item_stride + (i & 1) * pred_stride);
      y = *(const uint4 *)(b + (i >> 1) * 16);
    } else {
      x = ((const uint4 *)a)[i];
      y = ((const uint4 *)b)[i];
    }
    u32 ra[4] = { x.x, x.y, x.z, x.w }, rb[4] = { y.x, y.y, y.z, y.w };
    costs[i] = satd4x4_regs(ra, rb);
  }
}

// ---------------------------------------------------------------------------
// Frame-level kernels: block pairs described by kvz_hip_block_pair inside two
// planes.  Addresses are unaligned and strided, the reference plane may be
// addressed outside the frame (edge replication, image.c:320-444 /
// ipol-generic.c:731-784).
// ---------------------------------------------------------------------------
struct plane_t {
  const u8 *p;
  u32 stride;
  int w, h;          // clamp extents; w == 0 => no clamping (coordinates are inside)
};

// 8 pixels of row y starting at column x (with replication when clamping and
// the segment is not fully inside); bytes beyond `n` valid pixels are zero.
__device__ __forceinline__ uint2 load_seg8(const plane_t &pl, int x, int y, int n)
{
  uint2 r;
  u8 v[8];
  if (pl.w == 0 || (x >= 0 && x + 8 <= pl.w && y >= 0 && y < pl.h)) {
    const u8 *q = pl.p + (size_t)y * pl.stride + x;
    if (n == 8) {
      // unaligned 8-byte load assembled from bytes by the compiler when it cannot prove alignment
      __builtin_memcpy(&r, q, 8);
      return r;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (i < n) ? q[i] : (u8)0;
  } else {
    const int yy = clampi(y, 0, pl.h - 1);
    const u8 *row = pl.p + (size_t)yy * pl.stride;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (i < n) ? row[clampi(x + i, 0, pl.w - 1)] : (u8)0;
  }
  r.x = v[0] | (v[1] << 8) | (v[2] << 16) | ((u32)v[3] << 24);
  r.y = v[4] | (v[5] << 8) | (v[6] << 16) | ((u32)v[7] << 24);
  return r;
}

// reg_sad (picture-generic.c:86-99) / kvz_image_calc_sad (image.c:455-486) and
// pixels_calc_ssd (picture-generic.c:521-536).  `lanes` lanes (8, or the whole wave) share one block pair; a lane takes
// 8-pixel row segments round robin, so 8 neighbouring lanes read up to 64 contiguous bytes of a row.  Four segments
// per lane are in flight (issued one at a time their latencies add up); segments past the end load nothing.
template <bool SSD>
__device__ __forceinline__ u32 pair_sad_accum(const plane_t &p1, const plane_t &p2, const kvz_hip_block_pair &d, int sub, int lanes)
{
  const int w = d.width, h = SSD ? d.width : d.height;
  const int spr = (w + 7) >> 3;
  const int nseg = spr * h;
  u32 acc = 0;
  for (int t0 = sub; t0 < nseg; t0 += 4 * lanes) {
    uint2 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = t0 + lanes * k;
      a[k] = make_uint2(0u, 0u); b[k] = a[k];
      if (t < nseg) {
        const int y = t / spr, sx = (t - y * spr) << 3;
        const int n = (w - sx) < 8 ? (w - sx) : 8;
        a[k] = load_seg8(p1, d.x1 + sx, d.y1 + y, n);
        b[k] = load_seg8(p2, d.x2 + sx, d.y2 + y, n);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (SSD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int e0 = (int)((a[k].x >> (8 * q)) & 255) - (int)((b[k].x >> (8 * q)) & 255);
          int e1 = (int)((a[k].y >> (8 * q)) & 255) - (int)((b[k].y >> (8 * q)) & 255);
          acc += (u32)(e0 * e0 + e1 * e1);
        }
      } else {
        acc = sad_dword(a[k].x, b[k].x, acc);
        acc = sad_dword(a[k].y, b[k].y, acc);
      }
    }
  }
  return acc;
}

// Large batches: 8 lanes per pair, grid-stride.
template <bool SSD>
__global__ __launch_bounds__(256) void pair_sad_kernel(plane_t p1, plane_t p2, const kvz_hip_block_pair *__restrict__ pairs,
                                                       size_t count, u32 *__restrict__ out)
{
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t ngroups = ((size_t)gridDim.x * blockDim.x) >> 3;
  const int sub = threadIdx.x & 7;
  const size_t count_up = (count + 7) & ~(size_t)7;
  for (size_t i = tid >> 3; i < count_up; i += ngroups) {
    u32 acc = 0;
    if (i < count) acc = pair_sad_accum<SSD>(p1, p2, pairs[i], sub, 8);
    acc = group_sum<8>(acc);
    if (i < count && sub == 0) out[i] = acc;
  }
}

// Frame-sized batches (one launch per frame: a few hundred 64x64 pairs cannot fill the chip 8 lanes at a time): ONE WAVE
// PER DESCRIPTOR is launched, and each group of 8 consecutive descriptors decides from its sizes how to use its 8 waves --
// all pairs of at least 1024 pixels: every wave takes one pair with all 64 lanes; otherwise the group's first wave takes
// the 8 pairs 8 lanes each and the other seven exit at once.  One launch, no size hint from the host: a second launch
// for the large pairs costs more than it saves on small-pair batches, which are launch bound at a few microseconds.
__device__ __forceinline__ bool pair_group_is_large(const kvz_hip_block_pair *__restrict__ pairs, size_t base, size_t count)
{
  bool large = true;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (base + k < count) large = large && (pairs[base + k].width * pairs[base + k].height >= 1024);   // uniform address: scalar loads
  return large;
}
template <bool SSD>
__global__ __launch_bounds__(256) void pair_sad_wave_kernel(plane_t p1, plane_t p2, const kvz_hip_block_pair *__restrict__ pairs,
                                                            size_t count, u32 *__restrict__ out)
{
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t base = wave & ~(size_t)7;
  const int j = (int)(wave & 7), lane = threadIdx.x & 63;
  if (base >= count) return;
  if (pair_group_is_large(pairs, base, count)) {
    const size_t i = base + j;
    if (i >= count) return;
    const kvz_hip_block_pair d = pairs[i];
    const u32 acc = group_sum<64>(pair_sad_accum<SSD>(p1, p2, d, lane, 64));
    if (lane == 0) out[i] = acc;
  } else {
    if (j != 0) return;
    const size_t i = base + (lane >> 3);
    u32 acc = 0;
    if (i < count) acc = pair_sad_accum<SSD>(p1, p2, pairs[i], lane & 7, 8);
    acc = group_sum<8>(acc);
    if (i < count && (lane & 7) == 0) out[i] = acc;
  }
}

// 8x8 sub-block of a pair at (ox, oy) inside the block
__device__ __forceinline__ u32 satd8x8_planes(const plane_t &p1, const plane_t &p2, const kvz_hip_block_pair &d, int ox, int oy)
{
  u32 ra[16], rb[16];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint2 a = load_seg8(p1, d.x1 + ox, d.y1 + oy + r, 8);
    uint2 b = load_seg8(p2, d.x2 + ox, d.y2 + oy + r, 8);
    ra[2 * r] = a.x; ra[2 * r + 1] = a.y; rb[2 * r] = b.x; rb[2 * r + 1] = b.y;
  }
  return satd8x8_regs(ra, rb);
}
__device__ __forceinline__ u32 satd4x4_planes(const plane_t &p1, const plane_t &p2, const kvz_hip_block_pair &d, int ox, int oy)
{
  u32 ra[4], rb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ra[r] = load_seg8(p1, d.x1 + ox, d.y1 + oy + r, 4).x;
    rb[r] = load_seg8(p2, d.x2 + ox, d.y2 + oy + r, 4).x;
  }
  return satd4x4_regs(ra, rb);
}

// SATD_ANY_SIZE (strategies-picture.h:62-100) / kvz_image_calc_satd
// (image.c:488-545).  `lanes` lanes share a pair and split its 8x8 sub-blocks (and
// the 4x4 ones of a leading 4-pixel column / row when w or h is not a multiple
// of 8).
__device__ __forceinline__ u32 pair_satd_accum(const plane_t &p1, const plane_t &p2, const kvz_hip_block_pair &d, int sub, int lanes)
{
  u32 acc = 0;
  int w = d.width, h = d.height, ox = 0, oy = 0;
  if (w & 7) {                                   // first 4-px column, full height
    for (int y = sub * 4; y < h; y += 4 * lanes) acc += satd4x4_planes(p1, p2, d, 0, y);
    ox = 4; w -= 4;
  }
  if (h & 7) {                                   // first 4-px row of the rest
    for (int x = sub * 4; x < w; x += 4 * lanes) acc += satd4x4_planes(p1, p2, d, ox + x, 0);
    oy = 4; h -= 4;
  }
  const int w8 = w >> 3, n8 = w8 * (h >> 3);
  for (int t = sub; t < n8; t += lanes) {
    const int by = t / w8, bx = t - by * w8;
    acc += satd8x8_planes(p1, p2, d, ox + bx * 8, oy + by * 8);
  }
  return acc;
}
// A workgroup takes 64 consecutive descriptors.  When they are all 8x8 pairs -- the frame-level grid of
// kvz_image_calc_satd callers -- wave 0 scores them one pair per LANE in registers and the other waves leave: with the
// general split (8 lanes per pair, each lane one 8x8 sub-block) an 8x8 pair keeps one lane of eight busy (0.85 TB/s of
// its own traffic, 665 vector instructions per 8 pairs).  Any other mix runs the general split, eight pairs per wave and
// round.  Threads per workgroup ("pair_satd_threads"; one 1080p frame of 8x8 / of 16x16 pairs): 128: 7.0 / 16.5 us,
// 256: 8.0 / 9.9, 512: 10.5 / 8.6; before: 23.6 / 8.0.
__global__ __launch_bounds__(512) void pair_satd_kernel(plane_t p1, plane_t p2, const kvz_hip_block_pair *__restrict__ pairs,
                                                        size_t count, u32 *__restrict__ out)
{
  const int lane = threadIdx.x & 63, sub = lane & 7;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t chunks = (count + 63) >> 6;
  for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const size_t i = (c << 6) + lane;
    const bool have = i < count;
    kvz_hip_block_pair d = { 0, 0, 0, 0, 8, 8 };
    if (have) d = pairs[i];
    if (__ballot(have && (d.width != 8 || d.height != 8)) == 0) {
      if (wv == 0 && have) out[i] = satd8x8_planes(p1, p2, d, 0, 0);
      continue;
    }
    // all 16x16 (the other frame-level grid): a pair has four 8x8 sub-blocks, so FOUR lanes per pair and sixteen pairs per wave
    // round keep every lane busy, where the general split leaves half of each pair's eight lanes idle (9.9 us per 1080p frame)
    if (__ballot(have && (d.width != 16 || d.height != 16)) == 0) {
      for (int r = wv; r < 4; r += (int)(blockDim.x >> 6)) {
        const size_t k = (c << 6) + r * 16 + (lane >> 2);
        u32 acc = 0;
        if (k < count) acc = pair_satd_accum(p1, p2, pairs[k], lane & 3, 4);
        acc = group_sum<4>(acc);
        if (k < count && (lane & 3) == 0) out[k] = acc;
      }
      continue;
    }
    for (int r = wv; r < 8; r += (int)(blockDim.x >> 6)) {
      const size_t k = (c << 6) + r * 8 + (lane >> 3);
      u32 acc = 0;
      if (k < count) acc = pair_satd_accum(p1, p2, pairs[k], sub, 8);
      acc = group_sum<8>(acc);
      if (k < count && sub == 0) out[k] = acc;
    }
  }
}
// one wave per descriptor, see pair_sad_wave_kernel
__global__ __launch_bounds__(256) void pair_satd_wave_kernel(plane_t p1, plane_t p2, const kvz_hip_block_pair *__restrict__ pairs,
                                                             size_t count, u32 *__restrict__ out)
{
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t base = wave & ~(size_t)7;
  const int j = (int)(wave & 7), lane = threadIdx.x & 63;
  if (base >= count) return;
  if (pair_group_is_large(pairs, base, count)) {
    const size_t i = base + j;
    if (i >= count) return;
    const kvz_hip_block_pair d = pairs[i];
    const u32 acc = group_sum<64>(pair_satd_accum(p1, p2, d, lane, 64));
    if (lane == 0) out[i] = acc;
  } else {
    if (j != 0) return;
    const size_t i = base + (lane >> 3);
    u32 acc = 0;
    if (i < count) acc = pair_satd_accum(p1, p2, pairs[i], lane & 7, 8);
    acc = group_sum<8>(acc);
    if (i < count && (lane & 7) == 0) out[i] = acc;
  }
}

// satd_any_size_quad (picture-generic.c:392-456): 4 candidate blocks against
// one original.  Reproduces the reference exactly, including for sizes that are
// not multiples of 8: the 4x4 stages contribute nothing and only shrink
// w/h by 4, the 8x8 grid then starts at the block ORIGIN.  8 lanes per item,
// each lane evaluates its 8x8 positions for all four candidates so the
// original's rows are loaded once.
__global__ __launch_bounds__(256) void quad_satd_kernel(const u8 *__restrict__ preds, u32 pred_stride, size_t pred_item_stride,
                                                        plane_t orig, const kvz_hip_block_pair *__restrict__ pairs,
                                                        size_t count, u32 *__restrict__ out)
{
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t ngroups = ((size_t)gridDim.x * blockDim.x) >> 3;
  const int sub = threadIdx.x & 7;
  const size_t count_up = (count + 7) & ~(size_t)7;
  for (size_t i = tid >> 3; i < count_up; i += ngroups) {
    u32 acc[4] = { 0, 0, 0, 0 };
    if (i < count) {
      const kvz_hip_block_pair d = pairs[i];
      int w = d.width, h = d.height;
      if (w & 7) w -= 4;
      if (h & 7) h -= 4;
      const int w8 = (w + 7) >> 3, h8 = (h + 7) >> 3, n8 = (w > 0 && h > 0) ? w8 * h8 : 0;
      for (int t = sub; t < n8; t += 8) {
        const int by = t / w8, bx = t - by * w8;
        u32 ro[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          uint2 o = load_seg8(orig, d.x1 + bx * 8, d.y1 + by * 8 + r, 8);
          ro[2 * r] = o.x; ro[2 * r + 1] = o.y;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u8 *pp = preds + (i * 4 + k) * pred_item_stride + (size_t)(by * 8) * pred_stride + bx * 8;
          u32 rp[16];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            uint2 t2;
            __builtin_memcpy(&t2, pp + (size_t)r * pred_stride, 8);
            rp[2 * r] = t2.x; rp[2 * r + 1] = t2.y;
          }
          acc[k] += satd8x8_regs(ro, rp);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = group_sum<8>(acc[k]);
    if (i < count && sub < 4) {
      u32 v = acc[0];
      v = sub == 1 ? acc[1] : v; v = sub == 2 ? acc[2] : v; v = sub == 3 ? acc[3] : v;
      out[i * 4 + sub] = v;
    }
  }
}

// inter_recon_bipred blend (picture-generic.c:538-588): (s0 + s1 + 64) >> 7
// through the 32-bit clip; samples are held in int16 (pixels << 6).
template <bool HP0, bool HP1>
__global__ __launch_bounds__(256) void bipred_blend_kernel(const void *__restrict__ s0, const void *__restrict__ s1,
                                                           u8 *__restrict__ dst, size_t n)
{
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  for (size_t i = tid * 4; i < n; i += nthreads * 4) {
    u8 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i + k < n) {
        i16 a = HP0 ? ((const i16 *)s0)[i + k] : (i16)(((const u8 *)s0)[i + k] << 6);
        i16 b = HP1 ? ((const i16 *)s1)[i + k] : (i16)(((const u8 *)s1)[i + k] << 6);
        o[k] = fast_clip32(((i32)a + (i32)b + 64) >> 7);
      }
    }
    if (i + 3 < n) *(u32 *)(dst + i) = o[0] | (o[1] << 8) | (o[2] << 16) | ((u32)o[3] << 24);
    else for (int k = 0; k < 4 && i + k < n; ++k) dst[i + k] = o[k];
  }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

template <bool DUAL>
static int launch_sad(int n, const u8 *a, const u8 *b, size_t count, u32 *costs, size_t ps, size_t is, hipStream_t st)
{
  // count here = number of block pairs (2 * items for DUAL)
  const unsigned threads = 256;
#define KVZ_SAD_CASE(N, U)                                                                        \
  case N: {                                                                                       \
    size_t chunks = count * (N * N / 16);                                                         \
    unsigned grid = stream_grid(chunks, (threads / 64) * 64 * U, (unsigned)tuning("sad_wgs_per_cu", 256)); \
    hipLaunchKernelGGL((sad_nxn_kernel<N, U, DUAL>), dim3(grid), dim3(threads), 0, st, a, b, costs, count, ps, is); \
  } break;
  switch (n) {
    KVZ_SAD_CASE(4, 4)
    KVZ_SAD_CASE(8, 4)
    KVZ_SAD_CASE(16, 4)
    KVZ_SAD_CASE(32, 4)
    KVZ_SAD_CASE(64, 4)
    default: return kvzhip::invalid_arg("kvz_hip_sad_nxn_batch / kvz_hip_sad_nxn_dual_batch (n must be 4, 8, 16, 32 or 64)");
  }
#undef KVZ_SAD_CASE
  KVZ_CHECK_LAUNCH("sad_nxn_kernel");
  return KVZ_HIP_OK;
}

template <bool DUAL>
static int launch_satd(int n, const u8 *a, const u8 *b, size_t count, u32 *costs, size_t ps, size_t is, hipStream_t st)
{
  const unsigned threads = 256;
  switch (n) {
    case 4:
      if (!DUAL) hipLaunchKernelGGL((satd4_kernel<4>), dim3(stream_grid(count, threads * 4)), dim3(threads), 0, st, a, b, costs, count);
      else hipLaunchKernelGGL((satd_4x4_kernel<DUAL>), dim3(stream_grid(count, threads)), dim3(threads), 0, st, a, b, costs, count, ps, is);
      break;
    case 8:
      if (!DUAL) hipLaunchKernelGGL((satd8_kernel<4>), dim3(stream_grid(count * 4, threads * 4, (unsigned)tuning("satd8_wgs_per_cu", 256))), dim3(threads), 0, st, a, b, costs, count);
      else hipLaunchKernelGGL((satd_nxn_kernel<8, DUAL>), dim3(stream_grid(count, threads)), dim3(threads), 0, st, a, b, costs, count, ps, is);
      break;
    case 16:
      if (!DUAL) hipLaunchKernelGGL((satd16_kernel<4>), dim3(stream_grid(count * 16, threads * 4)), dim3(threads), 0, st, a, b, costs, count);
      else hipLaunchKernelGGL((satd_nxn_kernel<16, DUAL>), dim3(stream_grid(count * 4, threads)), dim3(threads), 0, st, a, b, costs, count, ps, is);
      break;
    case 32: hipLaunchKernelGGL((satd_nxn_kernel<32, DUAL>), dim3(stream_grid(count * 16, threads)), dim3(threads), 0, st, a, b, costs, count, ps, is); break;
    case 64: hipLaunchKernelGGL((satd_nxn_kernel<64, DUAL>), dim3(stream_grid(count * 64, threads)), dim3(threads), 0, st, a, b, costs, count, ps, is); break;
    default: return kvzhip::invalid_arg("kvz_hip_satd_nxn_batch / kvz_hip_satd_nxn_dual_batch (n must be 4, 8, 16, 32 or 64)");
  }
  KVZ_CHECK_LAUNCH("satd_nxn_kernel");
  return KVZ_HIP_OK;
}

// batches up to PAIR_WAVE_MAX descriptors (the large blocks of a frame) get one wave per descriptor; larger ones fill the chip anyway
constexpr size_t PAIR_WAVE_MAX = 4096;      // measured: at 32 400 16x16 pairs the seven idle waves per group double the time; at 1 920 64x64 pairs a wave each is 3.5x faster
template <bool SSD>
static void launch_pair_sad(const plane_t &p1, const plane_t &p2, const kvz_hip_block_pair *pairs, size_t count, u32 *out, hipStream_t st)
{
  if (count <= PAIR_WAVE_MAX && tuning("pair_wave_kernel", 1))
    hipLaunchKernelGGL((pair_sad_wave_kernel<SSD>), dim3((unsigned)((count + 3) / 4)), dim3(256), 0, st, p1, p2, pairs, count, out);
  else
    hipLaunchKernelGGL((pair_sad_kernel<SSD>), dim3(stream_grid(count, 32)), dim3(256), 0, st, p1, p2, pairs, count, out);
}

extern "C" {

int kvz_hip_sad_nxn_batch(int n, const kvz_hip_pixel *blk1, const kvz_hip_pixel *blk2, size_t count, uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!blk1 || !blk2 || !costs || !aligned16(blk1) || !aligned16(blk2)) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  return launch_sad<false>(n, blk1, blk2, count, costs, 0, 0, ctx_stream(s));
}

int kvz_hip_satd_nxn_batch(int n, const kvz_hip_pixel *blk1, const kvz_hip_pixel *blk2, size_t count, uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!blk1 || !blk2 || !costs || !aligned16(blk1) || !aligned16(blk2)) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  return launch_satd<false>(n, blk1, blk2, count, costs, 0, 0, ctx_stream(s));
}

int kvz_hip_sad_nxn_dual_batch(int n, const kvz_hip_pixel *preds, size_t pred_stride, size_t item_stride,
                               const kvz_hip_pixel *orig, size_t count, uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!preds || !orig || !costs || !aligned16(preds) || !aligned16(orig) || (pred_stride & 15) || (item_stride & 15)) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  return launch_sad<true>(n, preds, orig, count * 2, costs, pred_stride, item_stride, ctx_stream(s));
}

int kvz_hip_satd_nxn_dual_batch(int n, const kvz_hip_pixel *preds, size_t pred_stride, size_t item_stride,
                                const kvz_hip_pixel *orig, size_t count, uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!preds || !orig || !costs || !aligned16(preds) || !aligned16(orig) || (pred_stride & 15) || (item_stride & 15)) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  return launch_satd<true>(n, preds, orig, count * 2, costs, pred_stride, item_stride, ctx_stream(s));
}

int kvz_hip_reg_sad_batch(const kvz_hip_pixel *plane1, uint32_t stride1, const kvz_hip_pixel *plane2, uint32_t stride2,
                          const kvz_hip_block_pair *pairs, size_t count, uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!plane1 || !plane2 || !pairs || !costs) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  plane_t p1 = { plane1, stride1, 0, 0 }, p2 = { plane2, stride2, 0, 0 };
  launch_pair_sad<false>(p1, p2, pairs, count, costs, ctx_stream(s));
  KVZ_CHECK_LAUNCH("pair_sad_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_image_calc_sad_batch(const kvz_hip_pixel *pic, uint32_t pic_stride, const kvz_hip_pixel *ref, uint32_t ref_stride,
                                 int ref_w, int ref_h, const kvz_hip_block_pair *pairs, size_t count, uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!pic || !ref || !pairs || !costs || ref_w <= 0 || ref_h <= 0) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  plane_t p1 = { pic, pic_stride, 0, 0 }, p2 = { ref, ref_stride, ref_w, ref_h };
  launch_pair_sad<false>(p1, p2, pairs, count, costs, ctx_stream(s));
  KVZ_CHECK_LAUNCH("pair_sad_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_image_calc_satd_batch(const kvz_hip_pixel *pic, uint32_t pic_stride, const kvz_hip_pixel *ref, uint32_t ref_stride,
                                  int ref_w, int ref_h, const kvz_hip_block_pair *pairs, size_t count, uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!pic || !ref || !pairs || !costs || ref_w <= 0 || ref_h <= 0) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  plane_t p1 = { pic, pic_stride, 0, 0 }, p2 = { ref, ref_stride, ref_w, ref_h };
  if (count <= PAIR_WAVE_MAX && tuning("pair_wave_kernel", 1))
    hipLaunchKernelGGL(pair_satd_wave_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, ctx_stream(s), p1, p2, pairs, count, costs);
  else
  {
    int threads = tuning("pair_satd_threads", 256) & ~63;
    threads = threads < 64 ? 64 : (threads > 512 ? 512 : threads);
    hipLaunchKernelGGL(pair_satd_kernel, dim3(stream_grid(count, 64)), dim3((unsigned)threads), 0, ctx_stream(s), p1, p2, pairs, count, costs);
  }
  KVZ_CHECK_LAUNCH("pair_satd_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_pixels_calc_ssd_batch(const kvz_hip_pixel *plane1, uint32_t stride1, const kvz_hip_pixel *plane2, uint32_t stride2,
                                  const kvz_hip_block_pair *pairs, size_t count, uint32_t *ssd, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!plane1 || !plane2 || !pairs || !ssd) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  plane_t p1 = { plane1, stride1, 0, 0 }, p2 = { plane2, stride2, 0, 0 };
  launch_pair_sad<true>(p1, p2, pairs, count, ssd, ctx_stream(s));
  KVZ_CHECK_LAUNCH("pair_sad_kernel<ssd>");
  return KVZ_HIP_OK;
}

int kvz_hip_satd_any_size_quad_batch(const kvz_hip_pixel *preds, uint32_t pred_stride, size_t pred_item_stride,
                                     const kvz_hip_pixel *orig, uint32_t orig_stride, const kvz_hip_block_pair *pairs,
                                     size_t count, uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!preds || !orig || !pairs || !costs) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  plane_t po = { orig, orig_stride, 0, 0 };
  hipLaunchKernelGGL(quad_satd_kernel, dim3(stream_grid(count, 32)), dim3(256), 0, ctx_stream(s), preds, pred_stride, pred_item_stride, po, pairs, count, costs);
  KVZ_CHECK_LAUNCH("quad_satd_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_bipred_blend_batch(int w, int h, int hi_prec0, const void *src0, int hi_prec1, const void *src1,
                               kvz_hip_pixel *dst, size_t count, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!src0 || !src1 || !dst || w <= 0 || h <= 0) return kvzhip::invalid_arg(__func__);
  const size_t n = (size_t)w * h * count;
  if (n == 0) return KVZ_HIP_OK;
  const unsigned grid = stream_grid(n, 1024);
  hipStream_t st = ctx_stream(s);
  if (hi_prec0 && hi_prec1) hipLaunchKernelGGL((bipred_blend_kernel<true, true>), dim3(grid), dim3(256), 0, st, src0, src1, dst, n);
  else if (hi_prec0) hipLaunchKernelGGL((bipred_blend_kernel<true, false>), dim3(grid), dim3(256), 0, st, src0, src1, dst, n);
  else if (hi_prec1) hipLaunchKernelGGL((bipred_blend_kernel<false, true>), dim3(grid), dim3(256), 0, st, src0, src1, dst, n);
  else hipLaunchKernelGGL((bipred_blend_kernel<false, false>), dim3(grid), dim3(256), 0, st, src0, src1, dst, n);
  KVZ_CHECK_LAUNCH("bipred_blend_kernel");
  return KVZ_HIP_OK;
}

}  // extern "C"
