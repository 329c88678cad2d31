// reduce.h — small HBM-bound reductions.
#pragma once

namespace icl {

// part[(n * chunks + chunk)][c] = sum_{s in chunk} x[n*bstride + c*S + s]   (grid (chunks, C, N); the caller adds the rows of
// `part` in a fixed order with colsum_multi_kernel).  Reference: the bias gradient of nn.Conv3d (autograd of networks/utils.py:104,107).
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ part, long S, long bstride, long chunk) {
  const int c = blockIdx.y;
  const float* p = x + (long)blockIdx.z * bstride + (long)c * S;
  const long lo = (long)blockIdx.x * chunk;
  const long hi = (lo + chunk < S) ? lo + chunk : S;
  float s = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) s += p[i];
  __shared__ float red[4];
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) part[((long)blockIdx.z * gridDim.x + blockIdx.x) * gridDim.y + c] = s;
}

}  // namespace icl
