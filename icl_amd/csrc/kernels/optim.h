// optim.h — fused SGD with momentum and weight decay (torch.optim.SGD semantics, dampening 0, no Nesterov),
// the optimiser of every ICL trainer (/root/reference/code/train_inherent_consistent_unet_3D_BraTS.py:85-86,115):
//   d = g + wd * p;   m = first ? d : momentum * m + d;   p = p - lr * m
// One pass: reads p, g, m and writes p, m (20 B per parameter; 15.7 GB per step for the 785 M parameter model) —
// the ATen foreach path it replaces makes three passes.  HBM-bound, 16 B per lane.
#pragma once

namespace icl {

__device__ __forceinline__ void sgd_update4(float4& p, const float4& g, float4& m, float lr, float mom, float wd, int first) {
  float4 d = make_float4(g.x + wd * p.x, g.y + wd * p.y, g.z + wd * p.z, g.w + wd * p.w);
  if (first) m = d;
  else m = make_float4(mom * m.x + d.x, mom * m.y + d.y, mom * m.z + d.z, mom * m.w + d.w);
  p = make_float4(p.x - lr * m.x, p.y - lr * m.y, p.z - lr * m.z, p.w - lr * m.w);
}

__device__ __forceinline__ void sgd_range(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, long n, long start,
                                          long stride, float lr, float mom, float wd, int first) {
  const bool al = ((((unsigned long long)p) | ((unsigned long long)g) | ((unsigned long long)m)) & 15ull) == 0;
  if (al) {
    const long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    for (long i = start; i < n4; i += stride) {
      float4 pv = p4[i], mv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : m4[i];
      sgd_update4(pv, g4[i], mv, lr, mom, wd, first);
      p4[i] = pv;
      m4[i] = mv;
    }
    for (long i = (n4 << 2) + start; i < n; i += stride) {
      const float d = g[i] + wd * p[i];
      const float mv = first ? d : mom * m[i] + d;
      m[i] = mv;
      p[i] = p[i] - lr * mv;
    }
  } else {
    for (long i = start; i < n; i += stride) {
      const float d = g[i] + wd * p[i];
      const float mv = first ? d : mom * m[i] + d;
      m[i] = mv;
      p[i] = p[i] - lr * mv;
    }
  }
}

// lr_dev (optional): learning rate read from device memory, so a captured hipGraph follows the host's poly-LR schedule
__global__ __launch_bounds__(256) void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, long n,
                                                           float lr, float mom, float wd, int first, const float* __restrict__ lr_dev) {
  if (lr_dev) lr = *lr_dev;
  sgd_range(p, g, m, n, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x, lr, mom, wd, first);
}

constexpr int kSgdMulti = 48;
struct SgdMulti {
  float* p[kSgdMulti];
  const float* g[kSgdMulti];
  float* m[kSgdMulti];
  long n[kSgdMulti];
};

// grid (chunks, tensors): every small tensor gets gridDim.x workgroups
__global__ __launch_bounds__(256) void sgd_momentum_multi_kernel(SgdMulti t, float lr, float mom, float wd, int first,
                                                                 const float* __restrict__ lr_dev) {
  if (lr_dev) lr = *lr_dev;
  const int k = blockIdx.y;
  sgd_range(t.p[k], t.g[k], t.m[k], t.n[k], (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x, lr, mom, wd, first);
}

// ---- SGD step of a Linear weight whose gradient is still FACTORED:  dW[n][k] = sum_m g[m][n] * x[m][k]  (M <= a few hundred rows).
// The four 13,824 x 13,824 token-axis MLP weights of the aligners (Class_Decoder.mlp2, networks/unet_3D_icl.py:258,267) are
// 97 % of the model; their gradient is an outer product of M = batch*classes*heads rows.  Forming it costs a 764 MB write
// per matrix and the optimiser reads it straight back; data-parallel training would all-reduce 3 GB of it.  Here the
// update is applied directly from the factors: one pass over p and m (16 B per parameter instead of 4 + 20), and ranks
// exchange the factors (~1 MB) instead of the matrix.  x slices and g columns are staged in LDS in chunks of 32 rows.
constexpr int kSfRows = 16, kSfCols = 256, kSfChunk = 32;

// grid (ceil(K/256), ceil(N/16)), block 256: thread = float4 column kq (0..63) x row lane rl (rows 4*rl .. 4*rl+3).  K % 4 == 0.
__global__ __launch_bounds__(256) void sgd_factored_kernel(float* __restrict__ p, float* __restrict__ mom, const float* __restrict__ g,
                                                           const float* __restrict__ x, int M, int N, int K, float lr, float momentum,
                                                           float wd, int first, const float* __restrict__ lr_dev) {
  __shared__ float4 xs[kSfChunk][kSfCols / 4];
  __shared__ float gs[kSfChunk][kSfRows];
  if (lr_dev) lr = *lr_dev;
  const int n0 = blockIdx.y * kSfRows, k0 = blockIdx.x * kSfCols;
  const int kq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  float4 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int m0 = 0; m0 < M; m0 += kSfChunk) {
    __syncthreads();
    for (int it = threadIdx.x; it < kSfChunk * (kSfCols / 4); it += 256) {
      const int m = it / (kSfCols / 4), q = it % (kSfCols / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + m < M && k0 + q * 4 < K) v = *reinterpret_cast<const float4*>(x + (long)(m0 + m) * K + k0 + q * 4);
      xs[m][q] = v;
    }
    for (int it = threadIdx.x; it < kSfChunk * kSfRows; it += 256) {
      const int m = it / kSfRows, r = it % kSfRows;
      gs[m][r] = (m0 + m < M && n0 + r < N) ? g[(long)(m0 + m) * N + n0 + r] : 0.f;
    }
    __syncthreads();
    const int mc = (M - m0 < kSfChunk) ? M - m0 : kSfChunk;
    for (int m = 0; m < mc; ++m) {
      const float4 xv = xs[m][kq];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gv = gs[m][rl * 4 + r];
        acc[r].x += gv * xv.x; acc[r].y += gv * xv.y; acc[r].z += gv * xv.z; acc[r].w += gv * xv.w;
      }
    }
  }
  if (k0 + kq * 4 >= K) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + rl * 4 + r;
    if (n >= N) continue;
    const long idx = (long)n * K + k0 + kq * 4;
    float4 pv = *reinterpret_cast<float4*>(p + idx);
    float4 mv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(mom + idx);
    sgd_update4(pv, acc[r], mv, lr, momentum, wd, first);
    *reinterpret_cast<float4*>(p + idx) = pv;
    *reinterpret_cast<float4*>(mom + idx) = mv;
  }
}

}  // namespace icl
