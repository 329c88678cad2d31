// common.h — small device helpers shared by all kernels (wave = 64 lanes on gfx950).
// Written against the device environment (device_env_hip.h); includes nothing itself.
#pragma once

namespace icl {

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// Sum over a whole block of NT threads (NT multiple of 64, <= 1024); result valid in every thread.
// `red` is an LDS scratch array of at least NT/64 floats owned by the caller.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous use
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// relu(InstanceNorm(v)) from the (scale, shift) pair of norm_finalize_stats_kernel: what a consumer of a deferred normalisation sees
__device__ __forceinline__ float norm_relu1(float v, float sc, float sh) { return fmaxf(fmaf(v, sc, sh), 0.f); }
__device__ __forceinline__ float4 norm_relu4(float4 v, float sc, float sh) {
  return make_float4(norm_relu1(v.x, sc, sh), norm_relu1(v.y, sc, sh), norm_relu1(v.z, sc, sh), norm_relu1(v.w, sc, sh));
}

// component j of a float4 (j is a compile-time constant after unrolling)
__device__ __forceinline__ float f4c(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

__device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Chan et al. merge of two (count, mean, M2) summaries — numerically stable variance.
__device__ __forceinline__ void welford_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb;
  const float delta = meanb - mean;
  const float f = nb / nt;
  mean = mean + delta * f;
  m2 = m2 + m2b + delta * delta * n * f;
  n = nt;
}

}  // namespace icl
