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

}  // namespace icl
