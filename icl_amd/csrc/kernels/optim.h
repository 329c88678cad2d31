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
constexpr int kSvRows = 16, kSvCols = 256, kSvChunk = 32;

// Variant for M <= 32 factor rows (one rank): plain VALU FMAs, no transposition step.
// grid (ceil(K/256), ceil(N/16)), block 256: thread = float4 column kq (0..63) x row lane rl (rows 4*rl .. 4*rl+3).  K % 4 == 0.
// V: 0 = load p / m after the products, 1 = issue the p / m loads first (their HBM latency overlaps the staging and the FMAs),
//    2 = as 1 with non-temporal loads and stores (the default: p and m are touched once per step; 13,824^2, M = 8:
//    567 / 553 / 502 us = 5.4 / 5.5 / 6.1 TB/s, tools/sgd_probe.py).
template <int V>
__global__ __launch_bounds__(256) void sgd_factored_small_kernel(float* __restrict__ p, float* __restrict__ mom, const float* __restrict__ g,
                                                           const float* __restrict__ x, int M, int N, int K, float lr, float momentum,
                                                           float wd, int first, const float* __restrict__ lr_dev) {
  __shared__ float4 xs[kSvChunk][kSvCols / 4];
  __shared__ float gs[kSvChunk][kSvRows];
  if (lr_dev) lr = *lr_dev;
  const int n0 = blockIdx.y * kSvRows, k0 = blockIdx.x * kSvCols;
  const int kq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  float4 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pv[4], mv[4];
  const bool col_ok = k0 + kq * 4 < K;
  if (V >= 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + rl * 4 + r;
      pv[r] = mv[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col_ok && n < N) {
        const long idx = (long)n * K + k0 + kq * 4;
        pv[r] = V == 2 ? icl_nt_load4(p + idx) : *reinterpret_cast<const float4*>(p + idx);
        if (!first) mv[r] = V == 2 ? icl_nt_load4(mom + idx) : *reinterpret_cast<const float4*>(mom + idx);
      }
    }
  }
  for (int m0 = 0; m0 < M; m0 += kSvChunk) {
    __syncthreads();
    for (int it = threadIdx.x; it < kSvChunk * (kSvCols / 4); it += 256) {
      const int m = it / (kSvCols / 4), q = it % (kSvCols / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + m < M && k0 + q * 4 < K) v = *reinterpret_cast<const float4*>(x + (long)(m0 + m) * K + k0 + q * 4);
      xs[m][q] = v;
    }
    for (int it = threadIdx.x; it < kSvChunk * kSvRows; it += 256) {
      const int m = it / kSvRows, r = it % kSvRows;
      gs[m][r] = (m0 + m < M && n0 + r < N) ? g[(long)(m0 + m) * N + n0 + r] : 0.f;
    }
    __syncthreads();
    const int mc = (M - m0 < kSvChunk) ? M - m0 : kSvChunk;
    for (int m = 0; m < mc; ++m) {
      const float4 xv = xs[m][kq];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gv = gs[m][rl * 4 + r];
        acc[r].x += gv * xv.x; acc[r].y += gv * xv.y; acc[r].z += gv * xv.z; acc[r].w += gv * xv.w;
      }
    }
  }
  if (!col_ok) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + rl * 4 + r;
    if (n >= N) continue;
    const long idx = (long)n * K + k0 + kq * 4;
    if (V == 0) {
      pv[r] = *reinterpret_cast<float4*>(p + idx);
      mv[r] = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(mom + idx);
    }
    sgd_update4(pv[r], acc[r], mv[r], lr, momentum, wd, first);
    if (V == 2) {
      icl_nt_store4(p + idx, pv[r]);
      icl_nt_store4(mom + idx, mv[r]);
    } else {
      *reinterpret_cast<float4*>(p + idx) = pv[r];
      *reinterpret_cast<float4*>(mom + idx) = mv[r];
    }
  }
}

// The same update as a NARROW persistent launch (round 6): `gridDim.x` workgroups of 1,024 threads — one per CU, on as many CUs — walk the
// 16-row x 1,024-column tiles of the matrix; the four 256-thread quarters of a workgroup are four workgroups of the kernel above side by side
// (same thread-to-element map inside a 16 x 256 tile, same sums in the same order: bit-identical results).  What it is for: the 16 B / weight
// update streams of the two 13,824^2 matrices whose update is deferred out of their backward pass (FusedSGD.update_placement "deep") run
// UNDER the deep part of the backward pass (24^3 / 12^3 / 6^3 levels: launches of 72-144 workgroups that leave the HBM idle) instead of
// beside the aligners' serial query chain at the end of the forked phase.  The wide kernel cannot do that — its 46,656 short workgroups take
// every CU, and the deep levels' dependent launches queue behind them (round 4: +0.35 ms) —; a grid of ~100 fat workgroups holds ~100 CUs
// for the whole stream and leaves the others to the convolutions.  M <= 16 factor rows (one chunk), non-temporal loads and stores.
constexpr int kSnCols = 1024, kSnMaxRows = 16;
__global__ __launch_bounds__(1024) void sgd_factored_narrow_kernel(float* __restrict__ p, float* __restrict__ mom, const float* __restrict__ g,
                                                                   const float* __restrict__ x, int M, int N, int K, float lr, float momentum,
                                                                   float wd, int first, const float* __restrict__ lr_dev) {
  ICL_DYN_LDS(float4, lds4);                          // 65 KB: dynamic (a static array may not exceed 64 KB)
  float4 (*xs)[kSnCols / 4] = reinterpret_cast<float4 (*)[kSnCols / 4]>(lds4);                              // [kSnMaxRows][256] float4, 64 KB
  float (*gs)[kSvRows] = reinterpret_cast<float (*)[kSvRows]>(lds4 + kSnMaxRows * (kSnCols / 4));           // [kSnMaxRows][16]
  if (lr_dev) lr = *lr_dev;
  const int sub = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int kq = t & 63, rl = t >> 6;
  const int nbx = (K + kSnCols - 1) / kSnCols, nby = (N + kSvRows - 1) / kSvRows, ntiles = nbx * nby;
  // the p / m rows of the NEXT tile are requested before this tile's products: a workgroup always has one tile's 128 KB in flight while it
  // multiplies and stores another (without it a CU's 16 waves load, wait, multiply and store in lockstep)
  float4 pn[4], mn[4];
  auto request = [&](int tile, float4* pv, float4* mv) __attribute__((always_inline)) {
    const int by = tile / nbx, bx = tile - by * nbx;
    const int n0 = by * kSvRows, k0 = bx * kSnCols + sub * kSvCols;
    const bool ok = tile < ntiles && k0 + kq * 4 < K;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + rl * 4 + r;
      pv[r] = mv[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok && n < N) {
        const long idx = (long)n * K + k0 + kq * 4;
        pv[r] = icl_nt_load4(p + idx);
        if (!first) mv[r] = icl_nt_load4(mom + idx);
      }
    }
  };
  request(blockIdx.x, pn, mn);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int by = tile / nbx, bx = tile - by * nbx;
    const int n0 = by * kSvRows, k0 = bx * kSnCols + sub * kSvCols;
    const bool col_ok = k0 + kq * 4 < K;
    float4 pv[4], mv[4], acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pv[r] = pn[r];
      mv[r] = mn[r];
      acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    request(tile + gridDim.x, pn, mn);
    __syncthreads();      // the previous tile's readers of xs / gs are done
    for (int it = threadIdx.x; it < M * (kSnCols / 4); it += 1024) {
      const int m = it / (kSnCols / 4), q = it % (kSnCols / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bx * kSnCols + q * 4 < K) v = *reinterpret_cast<const float4*>(x + (long)m * K + bx * kSnCols + q * 4);
      xs[m][q] = v;
    }
    for (int it = threadIdx.x; it < M * kSvRows; it += 1024) {
      const int m = it / kSvRows, r = it % kSvRows;
      gs[m][r] = n0 + r < N ? g[(long)m * N + n0 + r] : 0.f;
    }
    __syncthreads();
    for (int m = 0; m < M; ++m) {
      const float4 xv = xs[m][sub * 64 + kq];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gv = gs[m][rl * 4 + r];
        acc[r].x += gv * xv.x; acc[r].y += gv * xv.y; acc[r].z += gv * xv.z; acc[r].w += gv * xv.w;
      }
    }
    if (col_ok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + rl * 4 + r;
        if (n >= N) continue;
        const long idx = (long)n * K + k0 + kq * 4;
        sgd_update4(pv[r], acc[r], mv[r], lr, momentum, wd, first);
        icl_nt_store4(p + idx, pv[r]);
        icl_nt_store4(mom + idx, mv[r]);
      }
    }
  }
}

constexpr int kSfRows = 64, kSfCols = 256, kSfChunk = 32;
constexpr int kSfXp = kSfCols + 16, kSfGp = kSfRows + 16;   // LDS row pitches: the 4 factor rows of an MFMA k-step land 16 banks apart
constexpr int kSfDp = kSfCols + 4;                            // pitch of the d tile: 16-byte aligned rows, conflict-free column writes
constexpr int kSfLdsFloats = kSfRows * kSfDp;                 // 64 x 260 floats = 66.6 KB (the operand slices alias its start)
static_assert(kSfChunk * (kSfXp + kSfGp) <= kSfLdsFloats, "operand slices must fit in the d-tile buffer");

// grid (ceil(K/256), ceil(N/64)), block 256 = 4 waves, dynamic LDS = kSfLdsFloats floats.  The workgroup owns the 64 x 256 block
// of the weight: wave w computes d = g^T x for columns 64 w .. 64 w + 63 on the fp32 MFMA (4 x 4 tiles, k-steps of 4 factor
// rows; A[n][m] = g[m][n], B[m][k] = x[m][k] from the LDS slices of the current 32-row chunk: 8 ds_read_b32 per 16 MFMAs),
// the block of d is transposed through LDS, and p / m are then updated with 16 bytes per lane and 1 KB per weight row and
// wave — the same access pattern as the dense SGD kernel, so the update stays an HBM stream until M reaches a few hundred
// (data-parallel ranks gather their factors: M = rows per rank x ranks).  K % 4 == 0.
__global__ __launch_bounds__(256) void sgd_factored_kernel(float* __restrict__ p, float* __restrict__ mom, const float* __restrict__ g,
                                                           const float* __restrict__ x, int M, int N, int K, float lr, float momentum,
                                                           float wd, int first, const float* __restrict__ lr_dev) {
  ICL_DYN_LDS(float, lds);
  float* xs = lds;
  float* gs = lds + kSfChunk * kSfXp;
  if (lr_dev) lr = *lr_dev;
  const int n0 = blockIdx.y * kSfRows, k0 = blockIdx.x * kSfCols;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr_ = lane & 15, lg = lane >> 4;
  const int wc = wid * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the factor slices of chunk c+1 are fetched into registers before the MFMAs of chunk c (8 float4 of x, 8 floats of g per thread)
  constexpr int XV = kSfChunk * (kSfCols / 4) / 256, GV = kSfChunk * kSfRows / 256;
  float4 xv[XV];
  float gv[GV];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int it = threadIdx.x + j * 256;
      const int m = it / (kSfCols / 4), q = it % (kSfCols / 4);
      // branch-free: out-of-range items load element 0 and are zeroed afterwards, so the loads issue back to back
      const bool ok = m0 + m < M && k0 + q * 4 < K;
      xv[j] = *reinterpret_cast<const float4*>(x + (ok ? (long)(m0 + m) * K + k0 + q * 4 : 0L));
      if (!ok) xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int it = threadIdx.x + j * 256;
      const int m = it / kSfRows, r = it % kSfRows;
      const bool ok = m0 + m < M && n0 + r < N;
      gv[j] = g[ok ? (long)(m0 + m) * N + n0 + r : 0L];
      if (!ok) gv[j] = 0.f;
    }
  };
  fetch(0);
  for (int m0 = 0; m0 < M; m0 += kSfChunk) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int it = threadIdx.x + j * 256;
      *reinterpret_cast<float4*>(xs + (it / (kSfCols / 4)) * kSfXp + (it % (kSfCols / 4)) * 4) = xv[j];
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int it = threadIdx.x + j * 256;
      gs[(it / kSfRows) * kSfGp + it % kSfRows] = gv[j];
    }
    __syncthreads();
    if (m0 + kSfChunk < M) fetch(m0 + kSfChunk);
    auto kstep = [&](int s) {
      float av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = gs[(s * 4 + lg) * kSfGp + a * 16 + lr_];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = xs[(s * 4 + lg) * kSfXp + wc + b * 16 + lr_];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = icl_mfma_16x16x4(av[a], bv[b], acc[a][b]);
    };
    if (M - m0 >= kSfChunk) {   // full chunk: straight-line code, operand reads of step s+1 under the MFMAs of step s
#pragma unroll
      for (int s = 0; s < kSfChunk / 4; ++s) kstep(s);
    } else {
      const int steps = (M - m0 + 3) / 4;   // rows past M are zero in LDS
      for (int s = 0; s < steps; ++s) kstep(s);
    }
  }
  // d block -> LDS (row-major), then the streaming update
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[(a * 16 + lg * 4 + r) * kSfDp + wc + b * 16 + lr_] = acc[a][b][r];
  __syncthreads();
  const int kq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  if (k0 + kq * 4 >= K) return;
  // eight rows of the thread in flight at once (16 x 16 B of loads per lane; sixteen would cost the second wave per SIMD): the
  // waves of this kernel spent 54 % of their life waiting on these loads (PMC) with only two workgroups per CU to cover for
  // each other, so fewer, deeper round trips
  constexpr int RB = 8;
#pragma unroll
  for (int j0 = 0; j0 < kSfRows / 4; j0 += RB) {
    float4 pv[RB], mv[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int n = n0 + rl + 4 * (j0 + j);
      pv[j] = mv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) {
        const long idx = (long)n * K + k0 + kq * 4;
        pv[j] = icl_nt_load4(p + idx);   // p and m are touched once per step: non-temporal, as in the small variant (+10 % there)
        if (!first) mv[j] = icl_nt_load4(mom + idx);
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int row = rl + 4 * (j0 + j), n = n0 + row;
      if (n >= N) continue;
      const long idx = (long)n * K + k0 + kq * 4;
      const float4 d = *reinterpret_cast<const float4*>(lds + row * kSfDp + kq * 4);
      sgd_update4(pv[j], d, mv[j], lr, momentum, wd, first);
      icl_nt_store4(p + idx, pv[j]);
      icl_nt_store4(mom + idx, mv[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ factored update, split products
// d = g^T x of the many-row case (gathered factors of a data-parallel step: M = rows per rank x ranks; nc = 16: 64 / 128 rows) is the
// fp32-MFMA part of sgd_factored_kernel and makes it compute-bound from ~64 rows (0.86 ms at M = 128 against 0.54 ms of pure
// streaming).  Here the products come from the bf16 matrix pipe with exact three-way splits of both factors (six MFMA terms per
// product, fp32 accumulation: fp32 accuracy, see kernels/conv_bf16x3.h): a 32-row chunk is ONE k = 32 MFMA step.
//
// factor_split_kernel: g [M][N], x [M][K] -> gs / xs [split 3][row block MB = ceil(M / 8)][column] of 8 packed bf16 (the 8 rows of the
// block for one column: the k-contiguous 16 bytes an MFMA lane group needs); rows >= M are zero.
__global__ __launch_bounds__(256) void factor_split_kernel(const float* __restrict__ g, const float* __restrict__ x, uint4* __restrict__ gs,
                                                           uint4* __restrict__ xs, int M, int N, int K) {
  const int MB = (M + 7) / 8;
  const long total = (long)MB * (N + K);
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int mb = (int)(it / (N + K)), c = (int)(it % (N + K));
    const bool isx = c >= N;
    const int col = isx ? c - N : c, ld = isx ? K : N;
    const float* src = (isx ? x : g) + col;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 8 * mb + j < M ? src[(long)(8 * mb + j) * ld] : 0.f;
    uint4 o1, o2, o3;
    bf3_split8(v, o1, o2, o3);
    uint4* d = (isx ? xs : gs) + (long)mb * ld + col;
    const long plane = (long)MB * ld;
    d[0] = o1;
    d[plane] = o2;
    d[2 * plane] = o3;
  }
}

// Same block structure, grid and update stream as sgd_factored_kernel; the operand slices of a 32-row chunk (4 row blocks x 3 splits:
// 48 KB of x, 12 KB of g) alias the d-tile buffer.  K % 4 == 0.
__global__ __launch_bounds__(256) void sgd_factored_split_kernel(float* __restrict__ p, float* __restrict__ mom, const uint4* __restrict__ gs,
                                                                 const uint4* __restrict__ xs, int M, int N, int K, float lr, float momentum,
                                                                 float wd, int first, const float* __restrict__ lr_dev) {
  ICL_DYN_LDS(float, lds);
  uint4* xl = reinterpret_cast<uint4*>(lds);              // [split 3][row block 4][column 256]
  uint4* gl = xl + 3 * 4 * kSfCols;                       // [split 3][row block 4][row n 64]
  static_assert((3 * 4 * kSfCols + 3 * 4 * kSfRows) * 16 <= kSfLdsFloats * 4, "operand slices must fit in the d-tile buffer");
  if (lr_dev) lr = *lr_dev;
  const int n0 = blockIdx.y * kSfRows, k0 = blockIdx.x * kSfCols;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr_ = lane & 15, lg = lane >> 4;
  const int wc = wid * 64;
  const int MB = (M + 7) / 8;
  const long xplane = (long)MB * K, gplane = (long)MB * N;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int XV = 3 * 4 * kSfCols / 256, GV = 3 * 4 * kSfRows / 256;      // 12 + 3 uint4 per thread and chunk
  uint4 xv[XV], gv[GV];
  auto fetch = [&](int mb0) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int it = threadIdx.x + j * 256, c = it % kSfCols, sb = it / kSfCols, blk = sb % 4, s = sb / 4;
      const bool ok = mb0 + blk < MB && k0 + c < K;
      xv[j] = xs[ok ? s * xplane + (long)(mb0 + blk) * K + k0 + c : 0L];
      if (!ok) xv[j] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int it = threadIdx.x + j * 256, r = it % kSfRows, sb = it / kSfRows, blk = sb % 4, s = sb / 4;
      const bool ok = mb0 + blk < MB && n0 + r < N;
      gv[j] = gs[ok ? s * gplane + (long)(mb0 + blk) * N + n0 + r : 0L];
      if (!ok) gv[j] = make_uint4(0u, 0u, 0u, 0u);
    }
  };
  fetch(0);
  for (int mb0 = 0; mb0 < MB; mb0 += 4) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < XV; ++j) xl[threadIdx.x + j * 256] = xv[j];
#pragma unroll
    for (int j = 0; j < GV; ++j) gl[threadIdx.x + j * 256] = gv[j];
    __syncthreads();
    if (mb0 + 4 < MB) fetch(mb0 + 4);
    // one MFMA k-step: lane group lg = row block of the chunk; A[n][rows] = g, B[rows][k] = x
    uint4 af[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int s = 0; s < 3; ++s) af[a][s] = gl[(s * 4 + lg) * kSfRows + a * 16 + lr_];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      uint4 bf[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) bf[s] = xl[(s * 4 + lg) * kSfCols + wc + b * 16 + lr_];
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        constexpr int sa[6] = {2, 1, 0, 1, 0, 0}, sb[6] = {0, 1, 2, 0, 1, 0};     // smallest terms first
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = icl_mfma_16x16x32_bf16(af[a][sa[t]], bf[sb[t]], acc[a][b]);
      }
    }
  }
  // d block -> LDS (row-major), then the streaming update (as sgd_factored_kernel)
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[(a * 16 + lg * 4 + r) * kSfDp + wc + b * 16 + lr_] = acc[a][b][r];
  __syncthreads();
  const int kq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  if (k0 + kq * 4 >= K) return;
  constexpr int RB = 8;
#pragma unroll
  for (int j0 = 0; j0 < kSfRows / 4; j0 += RB) {
    float4 pv[RB], mv[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int n = n0 + rl + 4 * (j0 + j);
      pv[j] = mv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) {
        const long idx = (long)n * K + k0 + kq * 4;
        pv[j] = icl_nt_load4(p + idx);
        if (!first) mv[j] = icl_nt_load4(mom + idx);
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int row = rl + 4 * (j0 + j), n = n0 + row;
      if (n >= N) continue;
      const long idx = (long)n * K + k0 + kq * 4;
      const float4 d = *reinterpret_cast<const float4*>(lds + row * kSfDp + kq * 4);
      sgd_update4(pv[j], d, mv[j], lr, momentum, wd, first);
      icl_nt_store4(p + idx, pv[j]);
      icl_nt_store4(mom + idx, mv[j]);
    }
  }
}

}  // namespace icl
