// mfma_probe.hip -- checks the operand / accumulator lane maps of
// v_mfma_i32_32x32x32_i8 on gfx950 with exact integer data, including the
// "accumulator as the next MFMA's operand" chaining the DCT kernel relies on.
// Development tool.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __host__ inline int kappa(int h, int e) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// D = A * B with lane (r = lane & 31, h = lane >> 5) holding A[r][16h + e] and B[16h + e][r], e = 0..15
__global__ void probe(const int8_t* A, const int8_t* B, int* D, int* D2, const int8_t* C2)
{
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  union { i32x4 v; int8_t b[16]; } a, b, c2;
  for (int e = 0; e < 16; ++e) { a.b[e] = A[r * 32 + 16 * h + e]; b.b[e] = B[(16 * h + e) * 32 + r]; }
  i32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, acc, 0, 0, 0);
  // documented C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  for (int g = 0; g < 16; ++g) D[kappa(h, g) * 32 + r] = acc[g];
  // chain: use acc (values fit i8 by construction of the test) as the B operand of D2 = C2 * X,
  // where X = D: lane (col r) element e = X[kappa(h,e)][r] = acc[e]; A = C2 with element e = C2[row r][kappa(h,e)]
  union { i32x4 v; int8_t b[16]; } xb;
  for (int e = 0; e < 16; ++e) { xb.b[e] = (int8_t)acc[e]; c2.b[e] = C2[r * 32 + kappa(h, e)]; }
  i32x16 acc2 = {0};
  acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(c2.v, xb.v, acc2, 0, 0, 0);
  for (int g = 0; g < 16; ++g) D2[kappa(h, g) * 32 + r] = acc2[g];
}

int main()
{
  int8_t hA[1024], hB[1024], hC2[1024];
  int hD[1024], hD2[1024], rD[1024], rD2[1024];
  srand(7);
  // small values so that D fits in int8 for the chained product: A in {-1,0,1}, B in [-3,3]
  for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 3 - 1); hB[i] = (int8_t)(rand() % 7 - 3); hC2[i] = (int8_t)(rand() % 255 - 127); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 32 + j]; rD[i * 32 + j] = s; }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += hC2[i * 32 + k] * (int8_t)rD[k * 32 + j]; rD2[i * 32 + j] = s; }
  int8_t *dA, *dB, *dC2; int *dD, *dD2;
  CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC2, 1024)); CK(hipMalloc(&dD, 4096)); CK(hipMalloc(&dD2, 4096));
  CK(hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dC2, hC2, 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dD2, dC2);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost)); CK(hipMemcpy(hD2, dD2, 4096, hipMemcpyDeviceToHost));
  int bad = 0, bad2 = 0, maxabs = 0;
  for (int i = 0; i < 1024; ++i) { bad += hD[i] != rD[i]; bad2 += hD2[i] != rD2[i]; if (abs(rD[i]) > maxabs) maxabs = abs(rD[i]); }
  printf("mfma_i32_32x32x32_i8: D mismatches %d / 1024 (max |D| %d), chained D2 mismatches %d / 1024\n", bad, maxabs, bad2);
  return bad || bad2;
}
