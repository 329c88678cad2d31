// token.h — aligner token operators: LayerNorm, GELU(erf) and the prototype cross-attention
// (/root/reference/code/networks/unet_3D_icl.py: Class_Decoder :244-268, Query_Attention :270-297, MLP :299-315).
// All of them are tiny next to the convolutions and the 13,824^2 mlp2 GEMMs: HBM/latency-bound, fp32 VALU.
// (The QK^T contraction is nc x 16 per token, <= 28 MFLOP per call — SURVEY.md §0.4 — far too small to feed MFMA.)
#pragma once

namespace icl {

// ---------------------------------------------------------------- LayerNorm over the last axis
// BLOCKROW = false: one wave per row (token rows, C = 64..256); true: one workgroup per row (the 13,824-long rows of norm3)
template <bool BLOCKROW>
__device__ __forceinline__ float team_sum(float v, float* red) {
  if (BLOCKROW) return block_sum<256>(v, red);
  return wave_sum(v);
}

template <bool BLOCKROW>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, long rows, int C, float eps) {
  __shared__ float red[4];
  const int lane = BLOCKROW ? threadIdx.x : (threadIdx.x & 63);
  const int stride = BLOCKROW ? 256 : 64;
  const long r = BLOCKROW ? (long)blockIdx.x : (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + r * C;
  float s = 0.f;
  for (int c = lane; c < C; c += stride) s += xr[c];
  const float m = team_sum<BLOCKROW>(s, red) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += stride) { const float d = xr[c] - m; q += d * d; }
  const float rs = 1.0f / sqrtf(team_sum<BLOCKROW>(q, red) / (float)C + eps);
  float* yr = y + r * C;
  for (int c = lane; c < C; c += stride) yr[c] = gamma ? (xr[c] - m) * rs * gamma[c] + beta[c] : (xr[c] - m) * rs;
  if (lane == 0) { mean[r] = m; rstd[r] = rs; }
}

// gx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = gy*gamma
template <bool BLOCKROW>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ gx, long rows, int C) {
  __shared__ float red[4];
  const int lane = BLOCKROW ? threadIdx.x : (threadIdx.x & 63);
  const int stride = BLOCKROW ? 256 : 64;
  const long r = BLOCKROW ? (long)blockIdx.x : (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + r * C;
  const float* gr = gy + r * C;
  const float m = mean[r], rs = rstd[r];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += stride) {
    const float g = gamma ? gr[c] * gamma[c] : gr[c];
    s1 += g;
    s2 += g * (xr[c] - m) * rs;
  }
  s1 = team_sum<BLOCKROW>(s1, red) / (float)C;
  s2 = team_sum<BLOCKROW>(s2, red) / (float)C;
  float* o = gx + r * C;
  for (int c = lane; c < C; c += stride) o[c] = rs * ((gamma ? gr[c] * gamma[c] : gr[c]) - s1 - (xr[c] - m) * rs * s2);
}

// Long rows (one workgroup per row: norm3's 13,824 tokens, 4-32 rows) with the row held in REGISTERS: thread t owns columns t, t + 256, ...
// (the assignment and the summation order of layernorm_fwd/bwd_kernel<true>, so the results are bit-identical), all K = ceil(C / 256) loads of
// a thread are issued before the first is used.  The plain kernels walk the row three times with one dependent 4-byte load per iteration: 70 us
// for a 55 KB row; these take the latency of one round of loads plus the two block reductions.  C <= 256 K.
template <int K>
__global__ __launch_bounds__(256) void layernorm_fwd_longrow_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ y,
                                                                    float* __restrict__ mean, float* __restrict__ rstd, int C, float eps) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  const long r = blockIdx.x;
  const float* xr = x + r * C;
  float v[K], ga[K], be[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = t + 256 * k;
    v[k] = c < C ? xr[c] : 0.f;
    ga[k] = (gamma && c < C) ? gamma[c] : 1.f;
    be[k] = (beta && c < C) ? beta[c] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (t + 256 * k < C) s += v[k];
  const float m = block_sum<256>(s, red) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (t + 256 * k < C) { const float d = v[k] - m; q += d * d; }
  const float rs = 1.0f / sqrtf(block_sum<256>(q, red) / (float)C + eps);
  float* yr = y + r * C;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = t + 256 * k;
    if (c < C) yr[c] = gamma ? (v[k] - m) * rs * ga[k] + be[k] : (v[k] - m) * rs;
  }
  if (t == 0) { mean[r] = m; rstd[r] = rs; }
}

template <int K>
__global__ __launch_bounds__(256) void layernorm_bwd_longrow_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, float* __restrict__ gx, int C) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  const long r = blockIdx.x;
  const float* xr = x + r * C;
  const float* gr = gy + r * C;
  const float m = mean[r], rs = rstd[r];
  float g[K], xh[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = t + 256 * k;
    const float gv = c < C ? gr[c] : 0.f;
    g[k] = (gamma && c < C) ? gv * gamma[c] : gv;
    xh[k] = c < C ? xr[c] : 0.f;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (t + 256 * k < C) {
      s1 += g[k];
      s2 += g[k] * (xh[k] - m) * rs;
    }
  s1 = block_sum<256>(s1, red) / (float)C;
  s2 = block_sum<256>(s2, red) / (float)C;
  float* o = gx + r * C;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = t + 256 * k;
    if (c < C) o[c] = rs * (g[k] - s1 - (xh[k] - m) * rs * s2);
  }
}

// Input gradient AND the row-chunk partials of dgamma / dbeta in one pass over gy and x (rows of fewer than 1024 columns, one wave per row): one
// launch instead of two on the backward chains of the aligner heads, where every dependent launch costs 10-15 us (profiles/r4_timeline.md).
// Workgroup j owns rows [j rpb, (j + 1) rpb); its four waves walk them RU rows at a time (RU independent rows in flight per wave), every lane
// keeps the running sums of its CPL columns, and the four waves' sums are folded through LDS in wave order into part_g[j][c], part_b[j][c] — the
// layout layernorm_wgrad_kernel writes, summed by colsum_multi_kernel in chunk order as before (no float atomics).  grid = chunks, C <= 64 CPL.
template <int CPL, int RU>
__global__ __launch_bounds__(256) void layernorm_bwd_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* __restrict__ gx,
                                                                  float* __restrict__ part_g, float* __restrict__ part_b, long rows, int C,
                                                                  long rpb) {
  __shared__ float fold[2][3][64 * CPL];
  const int lane = threadIdx.x & 63;
  int wid = threadIdx.x >> 6;
  ICL_WAVE_UNIFORM(wid);      // the row walk below is scalar control flow
  const long r0 = (long)blockIdx.x * rpb;
  const long r1 = r0 + rpb < rows ? r0 + rpb : rows;
  float a[CPL], b[CPL], gam[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    a[k] = b[k] = 0.f;
    gam[k] = lane + 64 * k < C ? gamma[lane + 64 * k] : 0.f;
  }
  for (long rb = r0 + wid * RU; rb < r1; rb += 4 * RU) {
    float gv[RU][CPL], xh[RU][CPL], rs[RU], s1[RU], s2[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const long r = rb + u < r1 ? rb + u : r1 - 1;            // a row past the chunk re-reads the last one; it is neither stored nor summed
      const float m = mean[r];
      rs[u] = rstd[r];
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        gv[u][k] = c < C ? gy[r * C + c] : 0.f;
        xh[u][k] = c < C ? x[r * C + c] - m : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      s1[u] = s2[u] = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        xh[u][k] *= rs[u];
        const float g = gv[u][k] * gam[k];
        s1[u] += g;
        s2[u] += g * xh[u][k];
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      s1[u] = wave_sum(s1[u]) / (float)C;
      s2[u] = wave_sum(s2[u]) / (float)C;
    }
    // b[k] is pinned in its own register after every add (ICL_PIN1: an empty asm, no instruction).  Round 5: without it the compiler
    // packs the two columns' `b += gv` into one v_pk_add_f32 whose halves are crossed (op_sel:[0,1] op_sel_hi:[1,0]) and whose
    // destination pair is also its second source; with kernels of other streams running beside this one (the aligner lanes) ONE row's
    // term then went missing from b[0] in lanes 48..63 in a few chunks of most launches — the same row's gx and its dgamma term, computed
    // from the same register, were exact (tests/diag/lane_dev_where.py; 21 of 30 steps with the packed add, 0 of 20 with the plain
    // v_add_f32; the wave-uniform row walk alone did not change it).  Mechanism not established; DESIGN.md section 8.
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (rb + u < r1) {
        float* o = gx + (rb + u) * C;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int c = lane + 64 * k;
          if (c < C) o[c] = rs[u] * (gv[u][k] * gam[k] - s1[u] - xh[u][k] * s2[u]);
          a[k] += gv[u][k] * xh[u][k];
          b[k] += gv[u][k];
          ICL_PIN1(b[k]);
        }
      }
    }
  }
  if (wid > 0) {
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      fold[0][wid - 1][lane + 64 * k] = a[k];
      fold[1][wid - 1][lane + 64 * k] = b[k];
    }
  }
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= C) continue;
      part_g[(long)blockIdx.x * C + c] = ((a[k] + fold[0][0][c]) + fold[0][1][c]) + fold[0][2][c];
      part_b[(long)blockIdx.x * C + c] = ((b[k] + fold[1][0][c]) + fold[1][1][c]) + fold[1][2][c];
    }
  }
}

// Partial sums of dgamma[c] = sum_rows gy*xhat and dbeta[c] = sum_rows gy: row chunk j (blockIdx.y) writes part_g[j][c] and
// part_b[j][c]; colsum_multi_kernel adds the chunks in a fixed order (no float atomics: the step is bit-reproducible).  A workgroup
// covers `cols` = min(C,256) columns x (256/cols) row lanes so that short rows still read full 256-byte lines; grid (ceil(C/cols), chunks).
__global__ __launch_bounds__(256) void layernorm_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              float* __restrict__ part_g, float* __restrict__ part_b, long rows, int C,
                                                              long rows_per_block, int cols) {
  const int lanes = 256 / cols;
  const int tc = threadIdx.x % cols;
  const int c = blockIdx.x * cols + tc;
  const int rl = threadIdx.x / cols;
  const bool valid = c < C && rl < lanes;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float a = 0.f, b = 0.f;
  if (valid) {
    // 8 rows in flight per thread: the loop is a pure stream of independent loads (one row every ~1 us otherwise)
    long r = r0 + rl;
    for (; r + 7L * lanes < r1; r += 8L * lanes) {
      float gv[8], xv[8], mu[8], rs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long rr = r + (long)u * lanes;
        gv[u] = gy[rr * C + c];
        xv[u] = x[rr * C + c];
        mu[u] = mean[rr];
        rs[u] = rstd[rr];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a += gv[u] * (xv[u] - mu[u]) * rs[u];
        b += gv[u];
      }
    }
    for (; r < r1; r += lanes) {
      const float g = gy[r * C + c];
      a += g * (x[r * C + c] - mean[r]) * rstd[r];
      b += g;
    }
  }
  // one partial per column and workgroup: fold the row lanes through LDS first
  __shared__ float sa[256], sb[256];
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  if (!valid || rl != 0) return;
  for (int l = 1; l < lanes; ++l) {
    a += sa[l * cols + tc];
    b += sb[l * cols + tc];
  }
  part_g[(long)blockIdx.y * C + c] = a;
  part_b[(long)blockIdx.y * C + c] = b;
}

// ---------------------------------------------------------------- GELU (exact erf form, nn.GELU default)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ gx, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * expf(-0.5f * v * v);
    gx[i] = gy[i] * (cdf + v * pdf);
  }
}

// ---------------------------------------------------------------- prototype cross-attention
// q    [B,h,nc,D]   (the reference's reshape-quirk layout of fc_q's output, unet_3D_icl.py:287)
// kv   [B,N,2,h,D]  (fc_kv output: k | v, head, d)
// logits[B,nc,h,N] = scale * q.k   — returned as the "attention" map (pre-softmax, :290,296), stored CLASS-major: the layout of the
// reference's `attn1.permute(0, 2, 1, 3)`, so the caller's permute is a no-op and the token-axis LayerNorm / MLP that follow read
// contiguous rows (as [B,h,nc,N] every consumer started with a transposing copy, forward and backward)
// out  [B,h,nc,D]   = softmax_N(logits) @ v;  stats[B,h,nc,2] = (row max, sum of exp) for the backward.
constexpr int kAttnMaxNc = 16;

template <int D>
__global__ __launch_bounds__(256) void attn_logits_kernel(const float* __restrict__ q, const float* __restrict__ kv, float* __restrict__ logits,
                                                          int B, int H, int nc, int N, float scale) {
  __shared__ float qs[kAttnMaxNc * D];
  const int bh = blockIdx.y;  // b*H + h
  const int b = bh / H, h = bh % H;
  for (int i = threadIdx.x; i < nc * D; i += 256) qs[i] = q[(long)bh * nc * D + i];
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* kp = kv + (((long)b * N + n) * 2 + 0) * H * D + h * D;
  float k[D];
#pragma unroll
  for (int d = 0; d < D; ++d) k[d] = kp[d];
  for (int c = 0; c < nc; ++c) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) s += qs[c * D + d] * k[d];
    logits[(((long)b * nc + c) * H + h) * N + n] = s * scale;
  }
}

// one workgroup per (b,h,c): softmax statistics and out = P @ V
// NT threads per row.  (Round 4: 1,024 threads per row for the 13,824-token rows of the 24^3 level — 8-16 rows per launch — measured SLOWER in the
// step's kernel statistics: softmax.V 41 us against 27, dQ 89 against 51-82; the launches use 256.)
template <int D, int NT>
__global__ __launch_bounds__(NT) void attn_softmax_pv_kernel(const float* __restrict__ logits, const float* __restrict__ kv,
                                                              float* __restrict__ out, float* __restrict__ stats, int B, int H, int nc, int N) {
  constexpr int NW = NT / 64;
  __shared__ float red[NW * (D + 1)];
  const int row = blockIdx.x;  // (b*H + h)*nc + c
  const int bh = row / nc;
  const int b = bh / H, h = bh % H;
  const float* l = logits + (((long)b * nc + row % nc) * H + h) * N;
  float m = -3.0e38f;
  for (int n = threadIdx.x; n < N; n += NT) m = fmaxf(m, l[n]);
  m = wave_max(m);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
  __syncthreads();
  float s = 0.f, o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = 0.f;
  for (int n = threadIdx.x; n < N; n += NT) {
    const float p = expf(l[n] - m);
    s += p;
    const float* vp = kv + (((long)b * N + n) * 2 + 1) * H * D + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] += p * vp[d];
  }
  s = wave_sum(s);
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = wave_sum(o[d]);
  if (lane == 0) {
    red[wid * (D + 1)] = s;
#pragma unroll
    for (int d = 0; d < D; ++d) red[wid * (D + 1) + 1 + d] = o[d];
  }
  __syncthreads();
  if (threadIdx.x < D + 1) {
    const int i = threadIdx.x;
    float v = red[i];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[w * (D + 1) + i];
    red[i] = v;
  }
  __syncthreads();
  const float tot = red[0];
  if (threadIdx.x < D) out[(long)row * D + threadIdx.x] = red[1 + threadIdx.x] / tot;
  if (threadIdx.x == 0) { stats[(long)row * 2] = m; stats[(long)row * 2 + 1] = tot; }
}

// dlogit[c][n] = ga[c][n] + p*(dO[c].V[n] - out[c].dO[c]);   thread per (b,h,n): dK[n], dV[n] (no reduction needed)
// dK, dV per token (one thread per token, 256 tokens per workgroup) and — `pq` given — the workgroup's share of dQ: pq[blockIdx.x][bh][c][d] =
// sum over its tokens of g[c][n] k[n][d] (attn_dq_reduce_kernel adds the workgroups' shares in order and scales).  With the token axis of the
// 24^3 level (13,824 tokens, 8-16 (sample, head, class) rows) the row-per-workgroup dQ kernel below is an 89 us latency chain on the query chain
// of the backward pass; here the products ride on the pass that has g and the token's K / V row in registers already.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ logits,
                                                          const float* __restrict__ stats, const float* __restrict__ out,
                                                          const float* __restrict__ gout, const float* __restrict__ glog,
                                                          float* __restrict__ gkv, int B, int H, int nc, int N, float scale,
                                                          float* __restrict__ pq) {
  __shared__ float qs[kAttnMaxNc * D], gos[kAttnMaxNc * D], dl[kAttnMaxNc], ms[kAttnMaxNc * 2];
  __shared__ float dqw[4][D];
  const int bh = blockIdx.y;
  const int b = bh / H, h = bh % H;
  for (int i = threadIdx.x; i < nc * D; i += 256) {
    qs[i] = q[(long)bh * nc * D + i];
    gos[i] = gout ? gout[(long)bh * nc * D + i] : 0.f;
  }
  for (int i = threadIdx.x; i < nc * 2; i += 256) ms[i] = stats[(long)bh * nc * 2 + i];
  __syncthreads();
  if (threadIdx.x < nc) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += out[((long)bh * nc + threadIdx.x) * D + d] * gos[threadIdx.x * D + d];
    dl[threadIdx.x] = s;  // delta_c
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool live = n < N;
  if (!live && pq == nullptr) return;
  const long base = ((long)b * N + (live ? n : N - 1)) * 2 * H * D + h * D;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float v[D], k[D], dk[D], dv[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    v[d] = kv[base + (long)H * D + d];
    k[d] = pq ? kv[base + d] : 0.f;
    dk[d] = 0.f; dv[d] = 0.f;
  }
  for (int c = 0; c < nc; ++c) {
    const long li = (((long)b * nc + c) * H + h) * N + (live ? n : N - 1);
    const float p = expf(logits[li] - ms[c * 2]) / ms[c * 2 + 1];
    float dp = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) dp += gos[c * D + d] * v[d];
    const float g = live ? (glog ? glog[li] : 0.f) + p * (dp - dl[c]) : 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { dk[d] += scale * g * qs[c * D + d]; dv[d] += p * gos[c * D + d]; }
    if (pq) {
      // the workgroup's sum of g k[d] for class c: wave sums, then the four waves in order
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float w = wave_sum(g * k[d]);
        if (lane == 0) dqw[wid][d] = w;
      }
      __syncthreads();
      if (threadIdx.x < D)
        pq[(((long)blockIdx.x * gridDim.y + bh) * nc + c) * D + threadIdx.x] =
            ((dqw[0][threadIdx.x] + dqw[1][threadIdx.x]) + dqw[2][threadIdx.x]) + dqw[3][threadIdx.x];
      __syncthreads();
    }
  }
  if (!live) return;
#pragma unroll
  for (int d = 0; d < D; ++d) { gkv[base + d] = dk[d]; gkv[base + (long)H * D + d] = dv[d]; }
}

// gq[row][d] = scale * sum over the token chunks x (in order) of pq[x][row][d]; rows = B H nc, one thread per (row, d)
__global__ __launch_bounds__(256) void attn_dq_reduce_kernel(const float* __restrict__ pq, float* __restrict__ gq, int chunks, int rd, float scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rd) return;
  float s = 0.f;
  int x = 0;
  for (; x + 8 <= chunks; x += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = pq[(long)(x + u) * rd + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; x < chunks; ++x) s += pq[(long)x * rd + i];
  gq[i] = scale * s;
}

// one workgroup per (b,h,c): dQ[c] = scale * sum_n dlogit[c][n] * K[n]
template <int D, int NT>
__global__ __launch_bounds__(NT) void attn_bwd_q_kernel(const float* __restrict__ kv, const float* __restrict__ logits, const float* __restrict__ stats,
                                                         const float* __restrict__ out, const float* __restrict__ gout,
                                                         const float* __restrict__ glog, float* __restrict__ gq, int B, int H, int nc,
                                                         int N, float scale) {
  constexpr int NW = NT / 64;
  __shared__ float red[NW * D];
  __shared__ float go[D];
  __shared__ float delta_s;
  const int row = blockIdx.x;
  const int bh = row / nc;
  const int b = bh / H, h = bh % H;
  if (threadIdx.x < D) go[threadIdx.x] = gout ? gout[(long)row * D + threadIdx.x] : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += out[(long)row * D + d] * go[d];
    delta_s = s;
  }
  __syncthreads();
  const float m = stats[(long)row * 2], tot = stats[(long)row * 2 + 1], delta = delta_s;
  const long lrow = (((long)b * nc + row % nc) * H + h) * N;      // class-major logits / glog row
  float acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;
  for (int n = threadIdx.x; n < N; n += NT) {
    const long base = ((long)b * N + n) * 2 * H * D + h * D;
    const float p = expf(logits[lrow + n] - m) / tot;
    float dp = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) dp += go[d] * kv[base + (long)H * D + d];
    const float g = (glog ? glog[lrow + n] : 0.f) + p * (dp - delta);
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] += g * kv[base + d];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float v = wave_sum(acc[d]);
    if (lane == 0) red[wid * D + d] = v;
  }
  __syncthreads();
  if (threadIdx.x < D) {
    const int d = threadIdx.x;
    float v = red[d];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[w * D + d];
    gq[(long)row * D + d] = scale * v;
  }
}

}  // namespace icl
