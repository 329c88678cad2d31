// norm.h — instance/batch normalisation (+ReLU) over NC[D]HW fp32 rows, forward and backward.
//
// Reference ops: nn.InstanceNorm3d(eps 1e-5, no affine) + ReLU inside UnetConv3
// (/root/reference/code/networks/utils.py:104-109) and nn.BatchNorm3d (batch statistics in
// train mode) + ReLU inside SeparableConv3d (networks/unet_3D_icl.py:317-345).
//
// A tensor is R = N*C rows of S contiguous floats.  Row r belongs to group r (instance norm)
// or r % C (batch norm).  All kernels are HBM-bound: rows are cut into chunks of kChunk
// elements so that even R = 16 rows (the 96^3 layers) fill the chip with workgroups.
//   forward : stats pass (read x) -> apply pass (merges the chunk summaries of its group, reads x, writes y)
//   backward: partial sums pass (read gy, x) -> apply (merges the partial sums of its group, reads gy, x, writes gx)
// Algorithmic HBM bytes per element: 12 B forward, 20 B backward.
#pragma once

namespace icl {

constexpr int kNormChunk = 8192;  // elements per workgroup
constexpr int kNormThreads = 256;

// (count, mean, M2) of xr[lo, hi) by the whole workgroup; the result is valid in thread 0
__device__ __forceinline__ void row_chunk_summary(const float* __restrict__ xr, long lo, long hi, long S, float& tn, float& tm, float& tq) {
  // shifted sums per thread (shift = first element seen) -> (n, mean, M2)
  float n = 0.f, shift = 0.f, s1 = 0.f, s2 = 0.f;
  if ((S & 3) == 0) {   // 16 bytes per lane (chunk bounds are multiples of 4)
    const float4* x4 = reinterpret_cast<const float4*>(xr);
#pragma unroll 4
    for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += kNormThreads) {
      const float4 v = x4[i];
      if (n == 0.f) shift = v.x;
      const float d0 = v.x - shift, d1 = v.y - shift, d2 = v.z - shift, d3 = v.w - shift;
      // the pair sums are pinned in registers of their own (ICL_PIN1: an empty asm, no instruction): left alone hipcc packs (a1, a2)
      // and (b1, b2) into register pairs and forms a + b as `v_pk_add_f32 d, d, d op_sel:[0,1] op_sel_hi:[1,0]` — the crossed form with
      // destination == source that lost a term in layernorm_bwd_wgrad_kernel under concurrent streams (round 5);
      // no kernel of the library keeps that form (tests/test_host_logic.py scans the device assembly).  Same sums, same order.
      float a1 = d0 + d1, b1 = d2 + d3, a2 = d0 * d0 + d1 * d1, b2 = d2 * d2 + d3 * d3;
      ICL_PIN1(a1); ICL_PIN1(b1); ICL_PIN1(a2); ICL_PIN1(b2);
      s1 += a1 + b1;
      s2 += a2 + b2;
      n += 4.f;
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += kNormThreads) {
      const float v = xr[i];
      if (n == 0.f) shift = v;
      const float d = v - shift;
      s1 += d;
      s2 += d * d;
      n += 1.f;
    }
  }
  float mean = 0.f, m2 = 0.f;
  if (n > 0.f) {
    mean = shift + s1 / n;
    m2 = s2 - s1 * s1 / n;
    if (m2 < 0.f) m2 = 0.f;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float nb = __shfl_xor(n, m, 64), mb = __shfl_xor(mean, m, 64), qb = __shfl_xor(m2, m, 64);
    // symmetric merge so every lane ends with the same value
    float na = n, ma = mean, qa = m2;
    welford_merge(na, ma, qa, nb, mb, qb);
    if (na > 0.f) { n = na; mean = ma; m2 = qa; }
  }
  __shared__ float red[3 * (kNormThreads / 64)];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[wid * 3] = n; red[wid * 3 + 1] = mean; red[wid * 3 + 2] = m2; }
  __syncthreads();
  tn = tm = tq = 0.f;
  if (threadIdx.x == 0) {
    tn = red[0]; tm = red[1]; tq = red[2];
    for (int w = 1; w < kNormThreads / 64; ++w) welford_merge(tn, tm, tq, red[w * 3], red[w * 3 + 1], red[w * 3 + 2]);
  }
}

// part[(r*nchunks + ch)*3 + {0,1,2}] = (count, mean, M2) of x[r, ch*chunk : (ch+1)*chunk]
__global__ __launch_bounds__(kNormThreads) void rowstats_partial_kernel(
    const float* __restrict__ x, float* __restrict__ part, long S, int nchunks) {
  const int ch = blockIdx.x;
  const long r = blockIdx.y;
  const long lo = (long)ch * kNormChunk;
  const long hi = (lo + kNormChunk < S) ? lo + kNormChunk : S;
  float tn, tm, tq;
  row_chunk_summary(x + r * S, lo, hi, S, tn, tm, tq);
  if (threadIdx.x == 0) {
    float* p = part + (r * nchunks + ch) * 3;
    p[0] = tn; p[1] = tm; p[2] = tq;
  }
}

// rstd from given variances (eval-mode BatchNorm): rstd[c] = 1/sqrt(var[c]+eps)
__global__ void rstd_from_var_kernel(const float* __restrict__ var, float* __restrict__ rstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) rstd[c] = 1.0f / sqrtf(var[c] + eps);
}

// Group statistics from the (count, mean, M2) chunk summaries, computed by EVERY workgroup that needs them (the summaries of a
// group are a few hundred floats): wave 0 merges them, LDS broadcasts the result.  Saves a separate finalize launch per
// normalisation layer (66 launches per U-Net step).  Returns (mean, M2/n, n) of group g in all threads.
__device__ __forceinline__ void norm_group_stats(const float* __restrict__ part, int g, int R, int C, int nchunks, int batch_mode,
                                                 float& mean_out, float& var_out, float& n_out) {
  __shared__ float bc[3];
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float n = 0.f, m = 0.f, q = 0.f;
    const int step = batch_mode ? C : R;
    const int nrows = (R - g + step - 1) / step;
    const long items = (long)nrows * nchunks;
    for (long it = lane; it < items; it += 64) {
      const int r = g + (int)(it / nchunks) * step;
      const float* p = part + ((long)r * nchunks + it % nchunks) * 3;
      welford_merge(n, m, q, p[0], p[1], p[2]);
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
      const float nb = __shfl_xor(n, sh, 64), mb = __shfl_xor(m, sh, 64), qb = __shfl_xor(q, sh, 64);
      welford_merge(n, m, q, nb, mb, qb);
    }
    if (lane == 0) { bc[0] = m; bc[1] = q / n; bc[2] = n; }
  }
  __syncthreads();
  mean_out = bc[0]; var_out = bc[1]; n_out = bc[2];
}

// Deferred normalisation (round 5): InstanceNorm statistics of a convolution output from the summaries its epilogue wrote — mean, rstd
// (for the backward pass) and the (scale, shift) = (rstd, -mean * rstd) pair with which the CONSUMERS of that output apply
// relu(fma(y, scale, shift)) while they load it (pool_resize.h maxpool2 / upsample2x / copy, misc.h conv1x1_stream_kernel, linear_wgrad.h
// conv1x1_wgrad_kernel): the normalised tensor of the 96^3 / 48^3 `conv2` layers is never written.  The same merge, in the same order, as
// norm_act_fwd_kernel's; the consumers' fma is the expression that kernel evaluates.  grid R (= samples x channels), 64 threads.
__global__ __launch_bounds__(64) void norm_finalize_stats_kernel(const float* __restrict__ part, int R, int C, int nchunks, float eps,
                                                                 float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ ss) {
  const int g = blockIdx.x;
  float m, var, n;
  norm_group_stats(part, g, R, C, nchunks, 0, m, var, n);
  if (threadIdx.x == 0) {
    const float rs = 1.0f / sqrtf(var + eps);
    mean[g] = m;
    rstd[g] = rs;
    ss[2 * g] = rs;
    ss[2 * g + 1] = 0.f - m * rs;
  }
}

// y = act(((x-mean[g])*rstd[g]) * gamma[c] + beta[c] [+ res]); act 0 = identity, 1 = ReLU, 2 = LeakyReLU(0.01)
// (nn.LeakyReLU default).  `res` (optional, same shape as x) is the residual branch of MONAI's UnetResBlock:
// lrelu(norm2(conv2(.)) + residual) in one pass.   grid (nchunks, R)
// part != nullptr: batch statistics — merged here from the chunk summaries; the first workgroup of a group stores mean / rstd
// for the backward pass and updates the BatchNorm running statistics.
__global__ __launch_bounds__(kNormThreads) void norm_act_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    long S, int C, int batch_mode, int act, const float* __restrict__ res, const float* __restrict__ part, int R, int nchunks,
    float eps, float* running_mean, float* running_var, float momentum, int inline_stats) {
  const long r = blockIdx.y;
  const int c = (int)(r % C);
  const int g = batch_mode ? c : (int)r;
  float m, rs;
  if (part == nullptr && inline_stats) {
    // instance statistics of a row that is ONE chunk: summarised here (the arithmetic of rowstats_partial_kernel, and merging a single
    // summary is exact), no statistics launch in front of this one
    __shared__ float bcs[2];
    float tn, tm, tq;
    row_chunk_summary(x + r * S, 0, S, S, tn, tm, tq);
    if (threadIdx.x == 0) { bcs[0] = tm; bcs[1] = tq / tn; }
    __syncthreads();
    m = bcs[0];
    rs = 1.0f / sqrtf(bcs[1] + eps);
    if (threadIdx.x == 0) { mean[g] = m; rstd[g] = rs; }
  } else if (part) {
    float var, n;
    norm_group_stats(part, g, R, C, nchunks, batch_mode, m, var, n);
    rs = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0 && blockIdx.x == 0 && (!batch_mode || r < C)) {
      mean[g] = m;
      rstd[g] = rs;
      if (running_mean) {
        const float unbiased = n > 1.f ? var * n / (n - 1.f) : var;
        running_mean[g] = (1.f - momentum) * running_mean[g] + momentum * m;
        running_var[g] = (1.f - momentum) * running_var[g] + momentum * unbiased;
      }
    }
  } else {
    m = mean[g];
    rs = rstd[g];
  }
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  const float sc = rs * ga, sh = be - m * rs * ga;
  const long lo = (long)blockIdx.x * kNormChunk;
  const long hi = (lo + kNormChunk < S) ? lo + kNormChunk : S;
  const float* xr = x + r * S;
  float* yr = y + r * S;
  const float* rr = res ? res + r * S : nullptr;
  if ((S & 3) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(xr);
    const float4* r4 = reinterpret_cast<const float4*>(rr);
    float4* y4 = reinterpret_cast<float4*>(yr);
#pragma unroll 4
    for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += kNormThreads) {
      float4 v = x4[i];
      v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
      if (rr) { const float4 q = r4[i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
      if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      else if (act == 2) { v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y; v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w; }
      y4[i] = v;
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += kNormThreads) {
      float v = xr[i] * sc + sh;
      if (rr) v += rr[i];
      yr[i] = act == 1 ? fmaxf(v, 0.f) : (act == 2 && v < 0.f ? 0.01f * v : v);
    }
  }
}

// Backward partial sums per (row, chunk): p1 = sum h, p2 = sum h*xhat, h = gy * [y > 0 if act].
// (sum h, sum h*xhat) over [lo, hi) of one row by the whole workgroup, valid in every thread
__device__ __forceinline__ void norm_bwd_chunk_sums(const float* __restrict__ xr, const float* __restrict__ gr, const float* __restrict__ rr, long lo,
                                                    long hi, long S, float m, float rs, float ga, float be, int act, float& p1, float& p2) {
  p1 = 0.f; p2 = 0.f;
  if ((S & 3) == 0) {   // 16 bytes per lane
    const float4* x4 = reinterpret_cast<const float4*>(xr);
    const float4* g4 = reinterpret_cast<const float4*>(gr);
    const float4* r4 = reinterpret_cast<const float4*>(rr);
#pragma unroll 4
    for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += kNormThreads) {
      const float4 xv = x4[i], gv = g4[i];
      const float4 rv = rr ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w}, rs4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (xs[k] - m) * rs;
        float h = gs[k];
        if (act && !(xh * ga + be + rs4[k] > 0.f)) h = (act == 2) ? 0.01f * h : 0.f;
        p1 += h;
        p2 += h * xh;
      }
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += kNormThreads) {
      const float xh = (xr[i] - m) * rs;
      float h = gr[i];
      if (act && !(xh * ga + be + (rr ? rr[i] : 0.f) > 0.f)) h = (act == 2) ? 0.01f * h : 0.f;
      p1 += h;
      p2 += h * xh;
    }
  }
  __shared__ float red[kNormThreads / 64];
  p1 = block_sum<kNormThreads>(p1, red);
  p2 = block_sum<kNormThreads>(p2, red);
}

__global__ __launch_bounds__(kNormThreads) void norm_act_bwd_partial_kernel(
    const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ part, long S, int C, int nchunks, int batch_mode, int act, const float* __restrict__ res) {
  const long r = blockIdx.y;
  const int c = (int)(r % C);
  const int g = batch_mode ? c : (int)r;
  const long lo = (long)blockIdx.x * kNormChunk;
  const long hi = (lo + kNormChunk < S) ? lo + kNormChunk : S;
  float p1, p2;
  norm_bwd_chunk_sums(x + r * S, gy + r * S, res ? res + r * S : nullptr, lo, hi, S, mean[g], rstd[g], gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f,
                      act, p1, p2);
  if (threadIdx.x == 0) {
    float* p = part + (r * nchunks + blockIdx.x) * 2;
    p[0] = p1; p[1] = p2;
  }
}

// gx = rstd*gamma*(h - (P1 + xhat*P2)/M)   (batch statistics)   or   rstd*gamma*h   (fixed statistics);  gres = h.
// P1 / P2 of the group are summed here from the per-(row, chunk) partials of norm_act_bwd_partial_kernel (every workgroup of the
// group does it: a few hundred floats; no separate reduction launch); the first workgroup of a channel also stores
// dgamma = P2, dbeta = P1 in batch mode.
__global__ __launch_bounds__(kNormThreads) void norm_act_bwd_apply_kernel(
    const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ part, float* __restrict__ gx, long S, int C, int batch_mode, int act,
    float inv_count, int use_batch_stats, const float* __restrict__ res, float* __restrict__ gres, int R, int nchunks,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const long r = blockIdx.y;
  const int c = (int)(r % C);
  const int g = batch_mode ? c : (int)r;
  __shared__ float bc[2];
  if (part == nullptr) {
    // instance statistics, the row is ONE chunk: its two sums computed here (norm_act_bwd_partial_kernel's arithmetic; with one chunk
    // the group sum below is that chunk's value), no partial-sum launch in front of this one
    float p1, p2;
    norm_bwd_chunk_sums(x + r * S, gy + r * S, res ? res + r * S : nullptr, 0, S, S, mean[g], rstd[g], gamma ? gamma[c] : 1.f,
                        beta ? beta[c] : 0.f, act, p1, p2);
    if (threadIdx.x == 0) { bc[0] = p1; bc[1] = p2; }
  } else if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int step = batch_mode ? C : R;
    const int nrows = (R - g + step - 1) / step;
    const long items = (long)nrows * nchunks;
    float p1 = 0.f, p2 = 0.f;
    for (long it = lane; it < items; it += 64) {
      const int rr_ = g + (int)(it / nchunks) * step;
      const float* p = part + ((long)rr_ * nchunks + it % nchunks) * 2;
      p1 += p[0];
      p2 += p[1];
    }
    p1 = wave_sum(p1);
    p2 = wave_sum(p2);
    if (lane == 0) { bc[0] = p1; bc[1] = p2; }
  }
  __syncthreads();
  const float P1 = bc[0], P2 = bc[1];
  if (batch_mode && dgamma && threadIdx.x == 0 && blockIdx.x == 0 && r < C) { dgamma[g] = P2; dbeta[g] = P1; }
  const float m = mean[g], rs = rstd[g];
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  const float a1 = use_batch_stats ? P1 * inv_count : 0.f;
  const float a2 = use_batch_stats ? P2 * inv_count : 0.f;
  const float k = rs * ga;
  const long lo = (long)blockIdx.x * kNormChunk;
  const long hi = (lo + kNormChunk < S) ? lo + kNormChunk : S;
  const float* xr = x + r * S;
  const float* gr = gy + r * S;
  float* o = gx + r * S;
  const float* rr = res ? res + r * S : nullptr;
  float* go = gres ? gres + r * S : nullptr;
  if ((S & 3) == 0) {   // 16 bytes per lane
    const float4* x4 = reinterpret_cast<const float4*>(xr);
    const float4* g4 = reinterpret_cast<const float4*>(gr);
    const float4* r4 = reinterpret_cast<const float4*>(rr);
    float4* o4 = reinterpret_cast<float4*>(o);
    float4* go4 = reinterpret_cast<float4*>(go);
#pragma unroll 4
    for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += kNormThreads) {
      const float4 xv = x4[i], gv = g4[i];
      const float4 rv = rr ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w}, rs4[4] = {rv.x, rv.y, rv.z, rv.w};
      float ov[4], hv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (xs[q] - m) * rs;
        float h = gs[q];
        if (act && !(xh * ga + be + rs4[q] > 0.f)) h = (act == 2) ? 0.01f * h : 0.f;
        ov[q] = k * (h - a1 - xh * a2);
        hv[q] = h;
      }
      o4[i] = make_float4(ov[0], ov[1], ov[2], ov[3]);
      if (go) go4[i] = make_float4(hv[0], hv[1], hv[2], hv[3]);
    }
    return;
  }
  for (long i = lo + threadIdx.x; i < hi; i += kNormThreads) {
    const float xh = (xr[i] - m) * rs;
    float h = gr[i];
    if (act && !(xh * ga + be + (rr ? rr[i] : 0.f) > 0.f)) h = (act == 2) ? 0.01f * h : 0.f;
    o[i] = k * (h - a1 - xh * a2);
    if (go) go[i] = h;
  }
}

}  // namespace icl
