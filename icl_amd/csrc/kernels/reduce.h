// reduce.h — small HBM-bound reductions.
#pragma once

namespace icl {

// out[c] += sum_{n, s in chunk} x[n*bstride + c*S + s]   (out pre-zeroed; grid (nchunks, C, N))
// Reference: the bias gradient of nn.Conv3d (autograd of networks/utils.py:104,107).
constexpr int kRedChunk = 16384;
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, long S, long bstride) {
  const int c = blockIdx.y;
  const float* p = x + (long)blockIdx.z * bstride + (long)c * S;
  const long lo = (long)blockIdx.x * kRedChunk;
  const long hi = (lo + kRedChunk < S) ? lo + kRedChunk : S;
  float s = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) s += p[i];
  __shared__ float red[4];
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) atomicAdd(out + c, s);
}

}  // namespace icl
