// How many contiguous bytes per weight row must one wave instruction cover to stream a 13,824^2 fp32 matrix at the HBM rate?
// Lane l reads 16 B; LPR lanes share a row (64/LPR rows per instruction, LPR*16 contiguous bytes each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ float4 ntload(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
// wave owns RW = 64/LPR rows... grid: blocks of 4 waves; wave w of block b owns rows (b*4+w)*RW .. +RW, walks all K
template <int LPR, int D>
__global__ __launch_bounds__(256) void sweep(const float* __restrict__ w, float* __restrict__ y, int K, int O) {
  constexpr int RW = 64 / LPR;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long row = ((long)blockIdx.x * 4 + wid) * RW + lane / LPR;
  const float* p = w + row * K + 4 * (lane % LPR);
  const int steps = K / (4 * LPR);
  float s = 0.f;
  for (int i = 0; i < steps; i += D) {
    float4 v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = (i + d < steps) ? ntload(p + (long)(i + d) * 4 * LPR) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int d = 0; d < D; ++d) s += v[d].x + v[d].y + v[d].z + v[d].w;
  }
  y[(long)blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void fill(float* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = (float)(i % 977) * 1e-3f;
}
template <typename F> float timeit(F f, int it = 10) {
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / it * 1e3f;
}
int main() {
  const int K = 13824, O = 13824;
  float *w, *y; CK(hipMalloc(&w, (size_t)K * O * 4)); CK(hipMalloc(&y, (size_t)O * 64 * 4 * 4));
  fill<<<4096, 256>>>(w, (long)K * O); CK(hipDeviceSynchronize());
  const double gb = (double)K * O * 4 / 1e9;
#define RUN(LPR, D) { float us = timeit([&] { sweep<LPR, D><<<O / (4 * (64 / LPR)), 256>>>(w, y, K, O); }); \
    printf("bytes/row/instr %4d  rows/instr %2d  D=%d  %8.1f us  %5.2f TB/s\n", LPR * 16, 64 / LPR, D, us, gb / us * 1e3); }
  RUN(64, 8) RUN(32, 8) RUN(16, 8) RUN(8, 8) RUN(4, 8) RUN(4, 16) RUN(16, 16) RUN(32, 16)
  return 0;
}
