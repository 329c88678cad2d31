// Micro-benchmarks behind linear_stream_nt_kernel (csrc/kernels/gemm.h): which part of the skinny forward product keeps the
// 13,824^2 weight stream below the HBM rate.  hipcc --offload-arch=gfx950 -O3 tools/probe/stream_nt_probe.hip -o /tmp/snt && /tmp/snt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float4 ntload(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 ldload(const float* p) { return *reinterpret_cast<const float4*>(p); }

// MODE 0: W fragment loads only (sum) | 1: + MFMA with constant x | 2: + x fragment loads from L2 (the real kernel)
// NT: non-temporal W loads.  CONTIG: each wave owns a contiguous K quarter instead of round-robin chunks.
template <int MODE, int D, bool NT, bool CONTIG>
__global__ __launch_bounds__(256) void frag_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int K, int O) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 15, lg = lane >> 4;
  const int o0 = blockIdx.x * 16;
  const float* wrow = w + (long)(o0 + r) * K + 4 * lg;
  const float* xrow = x + (long)r * K + 4 * lg;
  const int nchunks = K / 64;
  const int per = nchunks / 4;
  float4 wv[D][4], xv[D][4];
  auto chunk_of = [&](int it) { return CONTIG ? wid * per + it : wid + 4 * it; };
  auto load = [&](int d, int it) {
    const int c = chunk_of(it);
    const bool ok = it < per;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long off = ok ? 64 * c + 16 * j : 0;
      wv[d][j] = NT ? ntload(wrow + off) : ldload(wrow + off);
      if (MODE == 2) xv[d][j] = ldload(xrow + off);
    }
  };
  f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  float s = 0.f;
  auto compute = [&](int d) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) { s += wv[d][j].x + wv[d][j].y + wv[d][j].z + wv[d][j].w; }
      else {
        const float4 xx = MODE == 2 ? xv[d][j] : make_float4(1.f, 2.f, 3.f, 4.f);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.x, wv[d][j].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.y, wv[d][j].y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.z, wv[d][j].z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.w, wv[d][j].w, acc[1], 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) load(d, d);
  for (int it = 0; it < per; it += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) { compute(d); load(d, it + D + d); }
  }
  float out = s;
#pragma unroll
  for (int q = 0; q < 4; ++q) out += acc[0][q] + acc[1][q];
  y[(long)blockIdx.x * 256 + threadIdx.x] = out;
}

// ceiling: row-contiguous 16-byte loads (1 KiB per wave instruction), same grid/work split: block = 16 rows, wave = 4 rows
template <int D, bool NT>
__global__ __launch_bounds__(256) void rowread_kernel(const float* __restrict__ w, float* __restrict__ y, int K, int O) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float* base = w + (long)(blockIdx.x * 16 + wid * 4) * K + 4 * lane;
  float s = 0.f;
  const int steps = K / 256;
  for (int r = 0; r < 4; ++r) {
    const float* p = base + (long)r * K;
    for (int i = 0; i < steps; i += D) {
      float4 v[D];
#pragma unroll
      for (int d = 0; d < D; ++d) v[d] = (i + d < steps) ? (NT ? ntload(p + (long)(i + d) * 256) : ldload(p + (long)(i + d) * 256)) : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int d = 0; d < D; ++d) s += v[d].x + v[d].y + v[d].z + v[d].w;
    }
  }
  y[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void fill(float* p, long n, unsigned seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((h & 0xffff) / 65536.f - 0.5f) * 0.02f;
  }
}

template <typename F>
float timeit(F f, int it = 10) {
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  for (int i = 0; i < it; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / it * 1e3f;
}

int main() {
  const int K = 13824, O = 13824;
  float *w, *x, *y;
  CK(hipMalloc(&w, (size_t)K * O * 4)); CK(hipMalloc(&x, (size_t)32 * K * 4)); CK(hipMalloc(&y, (size_t)O * 64 * 4));
  fill<<<4096, 256>>>(w, (long)K * O, 1); fill<<<256, 256>>>(x, 32L * K, 2);
  CK(hipDeviceSynchronize());
  const double gb = (double)K * O * 4 / 1e9;
  const dim3 grid(O / 16);
#define RUN(name, ...) { float us = timeit([&] { __VA_ARGS__; }); printf("%-44s %8.1f us  %5.2f TB/s\n", name, us, gb / us * 1e3); }
  RUN("rowread D=4 nt", (rowread_kernel<4, true><<<grid, 256>>>(w, y, K, O)));
  RUN("rowread D=8 nt", (rowread_kernel<8, true><<<grid, 256>>>(w, y, K, O)));
  RUN("rowread D=8 plain", (rowread_kernel<8, false><<<grid, 256>>>(w, y, K, O)));
  RUN("frag W only D=3 nt rr", (frag_kernel<0, 3, true, false><<<grid, 256>>>(x, w, y, K, O)));
  RUN("frag W only D=3 plain rr", (frag_kernel<0, 3, false, false><<<grid, 256>>>(x, w, y, K, O)));
  RUN("frag W only D=3 nt contig", (frag_kernel<0, 3, true, true><<<grid, 256>>>(x, w, y, K, O)));
  RUN("frag W only D=6 nt rr", (frag_kernel<0, 6, true, false><<<grid, 256>>>(x, w, y, K, O)));
  RUN("frag W + mfma const x D=3 nt rr", (frag_kernel<1, 3, true, false><<<grid, 256>>>(x, w, y, K, O)));
  RUN("frag W + mfma + x(L2) D=3 nt rr", (frag_kernel<2, 3, true, false><<<grid, 256>>>(x, w, y, K, O)));
  RUN("frag W + mfma + x(L2) D=3 plain rr", (frag_kernel<2, 3, false, false><<<grid, 256>>>(x, w, y, K, O)));
  RUN("frag W + mfma + x(L2) D=3 nt contig", (frag_kernel<2, 3, true, true><<<grid, 256>>>(x, w, y, K, O)));
  RUN("frag W + mfma + x(L2) D=2 nt rr", (frag_kernel<2, 2, true, false><<<grid, 256>>>(x, w, y, K, O)));
  RUN("frag W + mfma + x(L2) D=4 nt rr", (frag_kernel<2, 4, true, false><<<grid, 256>>>(x, w, y, K, O)));
  return 0;
}
