// bw_probe.hip -- HBM streaming-bandwidth probe for MI355X: what a read-dominated
// integer kernel can reach, by unroll / grid size / cache policy.  Development tool,
// not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o tools/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload(const uint4* p) { u32x4 v = __builtin_nontemporal_load((const u32x4*)p); return make_uint4(v.x, v.y, v.z, v.w); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, bool NT, int NARR>
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, unsigned* __restrict__ out, size_t n16)
{
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  const size_t lane = threadIdx.x & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  for (size_t base = wave * 64 * U; base < n16; base += nwaves * 64 * U) {
    uint4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t i = base + u * 64 + lane;
      if (i < n16) {
        if (NT) { x[u] = ntload(a + i); if (NARR > 1) y[u] = ntload(b + i); }
        else { x[u] = a[i]; if (NARR > 1) y[u] = b[i]; }
      } else { x[u] = make_uint4(0, 0, 0, 0); y[u] = x[u]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc += x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
      if (NARR > 1) acc += y[u].x ^ y[u].y ^ y[u].z ^ y[u].w;
    }
  }
  if (acc == 0x12345678u) out[tid] = acc;
}

// block-contiguous variant: each workgroup owns one contiguous span of the arrays
template <int U, int NARR>
__global__ __launch_bounds__(256) void read_span_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, unsigned* __restrict__ out, size_t n16)
{
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
  unsigned acc = 0;
  for (size_t base = lo + threadIdx.x; base < hi; base += 256 * U) {
    uint4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t i = base + u * 256;
      if (i < hi) { x[u] = a[i]; if (NARR > 1) y[u] = b[i]; } else { x[u] = make_uint4(0, 0, 0, 0); y[u] = x[u]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc += x[u].x ^ x[u].y ^ x[u].z ^ x[u].w;
      if (NARR > 1) acc += y[u].x ^ y[u].y ^ y[u].z ^ y[u].w;
    }
  }
  if (acc == 0x12345678u) out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int U, bool NTL = false, bool NTS = false>
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ a, uint4* __restrict__ o, size_t n16)
{
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lane = threadIdx.x & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  for (size_t base = wave * 64 * U; base < n16; base += nwaves * 64 * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t i = base + u * 64 + lane;
      if (i < n16) {
        uint4 v = NTL ? ntload(a + i) : a[i];
        if (NTS) { u32x4 w = { v.x, v.y, v.z, v.w }; __builtin_nontemporal_store(w, (u32x4*)(o + i)); } else o[i] = v;
      }
    }
  }
}

// the fused quantize_residual traffic: two byte planes in (reference, prediction), one byte plane (reconstruction) and one
// int16 plane (coefficients) out -- 2 bytes read, 3 written per pixel, nothing computed
template <int U, bool NTS>
__global__ __launch_bounds__(256) void qr_mix_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ rec,
                                                     uint4* __restrict__ coef, size_t n16)
{
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t lane = threadIdx.x & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  for (size_t base = wave * 64 * U; base < n16; base += nwaves * 64 * U) {
    uint4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * 64 + lane;
      if (i < n16) { x[u] = ntload(a + i); y[u] = ntload(b + i); } else { x[u] = make_uint4(0, 0, 0, 0); y[u] = x[u]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * 64 + lane;
      if (i < n16) {
        const uint4 r = make_uint4(x[u].x ^ y[u].x, x[u].y ^ y[u].y, x[u].z ^ y[u].z, x[u].w ^ y[u].w);
        const uint4 c0 = make_uint4(x[u].x + y[u].y, x[u].y, y[u].z, x[u].w), c1 = make_uint4(y[u].x, x[u].z + y[u].w, y[u].y, x[u].x);
        if (NTS) {
          u32x4 w = { r.x, r.y, r.z, r.w }, w0 = { c0.x, c0.y, c0.z, c0.w }, w1 = { c1.x, c1.y, c1.z, c1.w };
          __builtin_nontemporal_store(w, (u32x4*)(rec + i));
          __builtin_nontemporal_store(w0, (u32x4*)(coef + 2 * i)); __builtin_nontemporal_store(w1, (u32x4*)(coef + 2 * i + 1));
        } else { rec[i] = r; coef[2 * i] = c0; coef[2 * i + 1] = c1; }
      }
    }
  }
}

template <typename F>
float timeit(F f, int iters = 20)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv)
{
  if (argc > 1 && !strcmp(argv[1], "qrmix")) {
    // 256 MiB per byte plane (what one launch of the fused kernels moves in bench_all: 268 MB read, 457 MB written incl. flags)
    const size_t bytes = (size_t)256 << 20, n16 = bytes / 16;
    uint4 *a, *b, *rec, *coef;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&rec, bytes)); CK(hipMalloc(&coef, 2 * bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    int grids[] = { 256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 128 };
    printf("%-34s %8s %10s\n", "variant (2 B read + 3 B written per pixel)", "grid", "GB/s");
#define RUN_MIX(U, NTS) for (int g : grids) { \
    float ms = timeit([&] { hipLaunchKernelGGL((qr_mix_kernel<U, NTS>), dim3(g), dim3(256), 0, 0, a, b, rec, coef, n16); }); \
    printf("qr mix U=%d nts=%d                   %8d %10.1f\n", U, (int)NTS, g, 5.0 * bytes / ms / 1e6); }
    RUN_MIX(1, false) RUN_MIX(2, false) RUN_MIX(4, false) RUN_MIX(1, true) RUN_MIX(2, true) RUN_MIX(4, true)
    return 0;
  }
  const size_t bytes = (size_t)512 << 20;   // 512 MiB per array (> 256 MiB Infinity Cache)
  const size_t n16 = bytes / 16;
  uint4 *a, *b, *o; unsigned* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, bytes)); CK(hipMalloc(&out, 1 << 26));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  int grids[] = { 256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 128 };
  printf("%-34s %8s %10s\n", "variant", "grid", "GB/s");
#define RUN_READ(U, NT, NARR) for (int g : grids) { \
    float ms = timeit([&] { hipLaunchKernelGGL((read_kernel<U, NT, NARR>), dim3(g), dim3(256), 0, 0, a, b, out, n16); }); \
    printf("read U=%d nt=%d arrays=%d              %8d %10.1f\n", U, (int)NT, NARR, g, NARR * (double)bytes / ms / 1e6); }
  RUN_READ(1, false, 1) RUN_READ(2, false, 1) RUN_READ(4, false, 1) RUN_READ(8, false, 1)
  RUN_READ(4, true, 1) RUN_READ(8, true, 1)
  RUN_READ(2, false, 2) RUN_READ(4, false, 2) RUN_READ(8, false, 2) RUN_READ(4, true, 2) RUN_READ(8, true, 2)
#define RUN_SPAN(U, NARR) for (int g : grids) { \
    float ms = timeit([&] { hipLaunchKernelGGL((read_span_kernel<U, NARR>), dim3(g), dim3(256), 0, 0, a, b, out, n16); }); \
    printf("span U=%d arrays=%d                   %8d %10.1f\n", U, NARR, g, NARR * (double)bytes / ms / 1e6); }
  RUN_SPAN(4, 2) RUN_SPAN(8, 2)
#define RUN_COPY(U) for (int g : grids) { \
    float ms = timeit([&] { hipLaunchKernelGGL((copy_kernel<U>), dim3(g), dim3(256), 0, 0, a, o, n16); }); \
    printf("copy U=%d (read+write bytes)         %8d %10.1f\n", U, g, 2.0 * bytes / ms / 1e6); }
  RUN_COPY(1) RUN_COPY(4)
#define RUN_COPY_NT(U, NTL, NTS) for (int g : grids) { \
    float ms = timeit([&] { hipLaunchKernelGGL((copy_kernel<U, NTL, NTS>), dim3(g), dim3(256), 0, 0, a, o, n16); }); \
    printf("copy U=%d ntl=%d nts=%d              %8d %10.1f\n", U, (int)NTL, (int)NTS, g, 2.0 * bytes / ms / 1e6); }
  RUN_COPY_NT(4, true, false) RUN_COPY_NT(4, false, true) RUN_COPY_NT(4, true, true) RUN_COPY_NT(2, true, true)
  return 0;
}
