// qchain.h — the QUERY chain of the aligner's Class_Decoder as a handful of fused stages (round 6).
//
// Reference: /root/reference/code/networks/unet_3D_icl.py:258-264 (Class_Decoder.forward, the query half), :283-297 (Query_Attention:
// fc_q, the read-out softmax(QK^T)V -> proj), :299-315 (MLP), :197,220-222 (query_convs: the Conv1d that hands the query down a level).
// The tensors are [R = batch * classes <= 32 rows, C <= 256] (the hidden layer 4C <= 1024) and the chain LINKS the three resolution
// levels of `sspa`, so in the backward pass it is the serial tail of the step's forked phase: mirrored operator by operator it was
// ~22 dependent launches per level forward and ~25 backward, each a 5 us kernel behind a 5-10 us dependency (profiles/r4_timeline.md,
// VERDICT round 5 item 1).  Here one launch is
//        rows -> [row prologue, recomputed by every workgroup] -> slice of a product with one weight matrix -> [element epilogue] -> rows
// so that LayerNorm (forward and backward), GELU (forward and backward), bias, the drop-path residual forms and the batch broadcast of
// the guided query ride on the product they feed / follow: 5 stage launches + the two attention launches per level forward, 6 + the
// attention backward + ONE launch for every parameter gradient of the level backward (qc_wgrad_kernel: they are leaves).
// Every sum runs in a fixed order (bit-reproducible); the LayerNorm arithmetic is that of token.h's kernels, term by term.
// Workgroups never talk to each other: a stage whose prologue needs whole rows gets them from the previous launch.
#pragma once

namespace icl {

constexpr int kQcRB = 8;         // rows per pass (a pass re-reads the workgroup's weight slice; R <= 8 for nc = 2: one pass)
constexpr int kQcMaxK = 1024;    // longest contraction (the MLP's hidden layer at C = 256)

enum { QC_PRO_NONE = 0, QC_PRO_LN = 1, QC_PRO_GELU = 2, QC_PRO_LNBWD = 3, QC_PRO_SCALE = 4 };
enum { QC_EPI_NONE = 0, QC_EPI_DP1 = 1, QC_EPI_RES_DP = 2, QC_EPI_GELUBWD = 3, QC_EPI_ADDROWS = 4, QC_EPI_SUMB = 5 };

// Mirrors IclQcStage (include/icl_hip.h) field by field.
struct QcStage {
  const float* x;       // [x_rows][K] input rows (nullptr: zeros); row r of the stage reads x row r % x_rows (x_rows < R: the guided query
  int x_rows;           //   [1, nc, C] broadcast over the batch without an expand copy)
  const float* w;       // trans 0: [N][K] (y = xs W^T, a Linear's forward); trans 1: [K][N] (y = xs W, its input gradient); trans 2: unused
  const float* bias;    // [N] or nullptr
  float* y;             // [R][N] (QC_EPI_SUMB: [nc][N])
  int R, K, N, trans, nc, npw;      // nc: rows per sample (sample of row r = r / nc); npw: outputs per wave (trans 0)
  int pro, epi;
  // prologue operands.  LN: pa = gamma, pb = beta.  LNBWD: x = d(normalised rows), pa = xhat, pb = rstd, pc = gamma, pd = residual rows
  // added to the result (or nullptr), pro_dp = 1: the result is scaled by (1 + f0(sample)).  SCALE: rows scaled by f1(sample).
  const float* pa; const float* pb; const float* pc; const float* pd;
  int pro_dp;
  // side outputs of the prologue, written by workgroup 0: so0 = xhat [R][K] and so1 = rstd [R] (LN), so2 = the prologue's result rows
  float* so0; float* so1; float* so2;
  // epilogue operands.  RES_DP: ea = residual rows [R][N].  GELUBWD: ea = pre-activation rows [R][N].  ADDROWS: ea / eb = rows added to
  // rows [ea_r0, ea_r1) / [eb_r0, eb_r1) of the result (gradients that reach a half of the batch).
  const float* ea; const float* eb;
  int ea_r0, ea_r1, eb_r0, eb_r1;
  // the two drop-path sites of Class_Decoder (:264 q + dp(q), :266 q + dp(mlp)): factor f(sample) = keep ? scale : 0; an inactive site
  // (eval mode, p = 0) has thresh 0 and scale 1: f = 1
  unsigned dp_seed[2], dp_thresh[2];
  float dp_scale[2];
  const unsigned* dp_seed_dev;
  float eps;
};

__device__ __forceinline__ float qc_dp_factor(const QcStage& s, int site, int b) {
  unsigned seed = s.dp_seed[site];
  if (s.dp_seed_dev) seed = mix32(seed ^ mix32(*s.dp_seed_dev + 0x632BE5ABu));      // as dropout_kernel
  return drop_apply(1.0f, seed, s.dp_thresh[site], s.dp_scale[site], (long)b);
}

__device__ __forceinline__ float qc_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float qc_gelu_grad(float v) {
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}

// Rows [r0, r0 + nr) of the stage's input through the prologue into xs[row][k] (row pitch K).  All threads of the workgroup; one wave
// per row for the row-wise sums (lane-strided, the order of token.h's one-wave-per-row kernels).  Ends with a barrier.
__device__ __forceinline__ void qc_prologue(const QcStage& s, float* __restrict__ xs, int r0, int nr) {
  const int K = s.K, nthreads = blockDim.x, nwaves = nthreads >> 6, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool side = blockIdx.x == 0;
  const int xr_mod = s.x_rows > 0 ? s.x_rows : s.R;
  if (s.pro == QC_PRO_LN || s.pro == QC_PRO_LNBWD) {
    for (int row = wid; row < nr; row += nwaves) {
      const int r = r0 + row;
      const float* xr = s.x + (long)(r % xr_mod) * K;
      float* o = xs + row * K;
      if (s.pro == QC_PRO_LN) {
        float t = 0.f;
        for (int c = lane; c < K; c += 64) t += xr[c];
        const float m = wave_sum(t) / (float)K;
        float q = 0.f;
        for (int c = lane; c < K; c += 64) { const float d = xr[c] - m; q += d * d; }
        const float rs = 1.0f / sqrtf(wave_sum(q) / (float)K + s.eps);
        for (int c = lane; c < K; c += 64) {
          const float xh = (xr[c] - m) * rs;
          const float v = s.pa ? (xr[c] - m) * rs * s.pa[c] + s.pb[c] : xh;
          o[c] = v;
          if (side) {
            if (s.so0) s.so0[(long)r * K + c] = xh;
            if (s.so2) s.so2[(long)r * K + c] = v;
          }
        }
        if (side && lane == 0 && s.so1) s.so1[r] = rs;
      } else {
        // gx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = gy * gamma  (token.h layernorm_bwd_kernel)
        const float* xh = s.pa + (long)r * K;
        const float rs = s.pb[r];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < K; c += 64) {
          const float g = s.pc ? xr[c] * s.pc[c] : xr[c];
          s1 += g;
          s2 += g * xh[c];
        }
        s1 = wave_sum(s1) / (float)K;
        s2 = wave_sum(s2) / (float)K;
        const float f = s.pro_dp ? 1.0f + qc_dp_factor(s, 0, r / s.nc) : 1.0f;
        for (int c = lane; c < K; c += 64) {
          float v = rs * ((s.pc ? xr[c] * s.pc[c] : xr[c]) - s1 - xh[c] * s2);
          if (s.pd) v += s.pd[(long)r * K + c];
          v *= f;
          o[c] = v;
          if (side && s.so2) s.so2[(long)r * K + c] = v;
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < nr * K; i += nthreads) {
      const int row = i / K, c = i - row * K, r = r0 + row;
      float v = s.x ? s.x[(long)(r % xr_mod) * K + c] : 0.f;
      if (s.pro == QC_PRO_GELU) v = qc_gelu(v);
      else if (s.pro == QC_PRO_SCALE) v *= qc_dp_factor(s, 1, r / s.nc);
      xs[i] = v;
      if (side && s.so2 && s.pro != QC_PRO_NONE) s.so2[(long)r * K + c] = v;
    }
  }
  __syncthreads();
}

// v = the product's value for (row r, output n)
__device__ __forceinline__ float qc_epilogue(const QcStage& s, float v, int r, int n) {
  if (s.bias) v += s.bias[n];
  switch (s.epi) {
    case QC_EPI_DP1: v *= 1.0f + qc_dp_factor(s, 0, r / s.nc); break;
    case QC_EPI_RES_DP: v = s.ea[(long)r * s.N + n] + qc_dp_factor(s, 1, r / s.nc) * v; break;
    case QC_EPI_GELUBWD: v *= qc_gelu_grad(s.ea[(long)r * s.N + n]); break;
    case QC_EPI_ADDROWS:
      if (s.ea && r >= s.ea_r0 && r < s.ea_r1) v += s.ea[(long)(r - s.ea_r0) * s.N + n];
      if (s.eb && r >= s.eb_r0 && r < s.eb_r1) v += s.eb[(long)(r - s.eb_r0) * s.N + n];
      break;
    default: break;
  }
  return v;
}

// trans 0: y[r][n] = epi(sum_k xs[r][k] W[n][k]).  256 threads; wave w of workgroup b owns outputs (4 b + w) npw .. + npw - 1; a lane's
// float4 of a weight row meets the same float4 of every row of xs (LDS, 16 consecutive bytes per lane: no conflicts).
__global__ __launch_bounds__(256) void qc_stage_fwd_kernel(QcStage s) {
  ICL_DYN_LDS(float, xs);
  const int K = s.K, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + wid) * s.npw;
  for (int r0 = 0; r0 < s.R; r0 += kQcRB) {
    const int nr = s.R - r0 < kQcRB ? s.R - r0 : kQcRB;
    if (r0) __syncthreads();
    qc_prologue(s, xs, r0, nr);
    // every weight load of the wave in flight before the first use
    float4 wv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int n = n0 + j, k = 4 * lane + 256 * p;
        wv[j][p] = (j < s.npw && n < s.N && k < K) ? *reinterpret_cast<const float4*>(s.w + (long)n * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j;
      if (j >= s.npw || n >= s.N) break;
      float acc[kQcRB];
#pragma unroll
      for (int r = 0; r < kQcRB; ++r) acc[r] = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int k = 4 * lane + 256 * p;
        if (k < K) {
#pragma unroll
          for (int r = 0; r < kQcRB; ++r)
            if (r < nr) {
              const float4 xv = *reinterpret_cast<const float4*>(xs + r * K + k);
              acc[r] += (wv[j][p].x * xv.x + wv[j][p].y * xv.y) + (wv[j][p].z * xv.z + wv[j][p].w * xv.w);
            }
        }
      }
      float mine = 0.f;
#pragma unroll
      for (int r = 0; r < kQcRB; ++r) {
        const float t = wave_sum(acc[r]);
        if (lane == r) mine = t;
      }
      if (lane < nr) s.y[(long)(r0 + lane) * s.N + n] = qc_epilogue(s, mine, r0 + lane, n);
    }
  }
}

// trans 1: y[r][c] = epi(sum_k xs[r][k] W[k][c]).  512 threads; a workgroup owns 64 output columns (256 contiguous bytes of every
// weight row); the 32 row groups (16 lanes x float4 each) take the weight rows round-robin, sixteen loads of a lane in flight together;
// partial sums meet in LDS and are added in wave order.  (Two waves per SIMD: 256 registers each.  A first version with 1,024 threads
// had 128, spilled 336 B per lane — and a kernel whose scratch demand over the whole chip exceeds the runtime's per-dispatch limit gets
// its scratch allocated and freed around EVERY launch: 70 us on average, 500 at worst, for a 9 us kernel.)
constexpr int kQcBwdWaves = 8;
__global__ __launch_bounds__(64 * kQcBwdWaves) void qc_stage_bwd_kernel(QcStage s) {
  ICL_DYN_LDS(float, lds);
  const int K = s.K, N = s.N;
  float* xs = lds;                            // [kQcRB][K]
  float* red = lds + kQcRB * kQcMaxK;         // [waves][kQcRB][64]
  constexpr int NRG = 4 * kQcBwdWaves;        // row groups
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane & 15, rg = wid * 4 + (lane >> 4);
  const int c0 = blockIdx.x * 64 + 4 * q;
  for (int r0 = 0; r0 < s.R; r0 += kQcRB) {
    const int nr = s.R - r0 < kQcRB ? s.R - r0 : kQcRB;
    if (r0) __syncthreads();
    qc_prologue(s, xs, r0, nr);
    float4 acc[kQcRB];
#pragma unroll
    for (int r = 0; r < kQcRB; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kb = 0; kb < K; kb += NRG * 16) {
      float4 wv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = kb + rg + NRG * i;
        wv[i] = (k < K && c0 < N) ? *reinterpret_cast<const float4*>(s.w + (long)k * N + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = kb + rg + NRG * i;
        if (k < K) {
#pragma unroll
          for (int r = 0; r < kQcRB; ++r)
            if (r < nr) {
              const float xv = xs[r * K + k];
              acc[r].x += xv * wv[i].x; acc[r].y += xv * wv[i].y; acc[r].z += xv * wv[i].z; acc[r].w += xv * wv[i].w;
            }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kQcRB; ++r) {      // the four row groups of a wave
      acc[r].x += __shfl_xor(acc[r].x, 16, 64); acc[r].y += __shfl_xor(acc[r].y, 16, 64);
      acc[r].z += __shfl_xor(acc[r].z, 16, 64); acc[r].w += __shfl_xor(acc[r].w, 16, 64);
      acc[r].x += __shfl_xor(acc[r].x, 32, 64); acc[r].y += __shfl_xor(acc[r].y, 32, 64);
      acc[r].z += __shfl_xor(acc[r].z, 32, 64); acc[r].w += __shfl_xor(acc[r].w, 32, 64);
      if (lane < 16 && r < nr) *reinterpret_cast<float4*>(red + (wid * kQcRB + r) * 64 + 4 * q) = acc[r];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nr * 64; t += 64 * kQcBwdWaves) {
      const int r = t >> 6, c = t & 63, n = blockIdx.x * 64 + c;
      if (n >= N) continue;
      float v = 0.f;
      for (int w = 0; w < kQcBwdWaves; ++w) v += red[(w * kQcRB + r) * 64 + c];
      s.y[(long)(r0 + r) * N + n] = qc_epilogue(s, v, r0 + r, n);
    }
  }
}

// trans 2: y = epi(xs) — a prologue with no product behind it (the LayerNorm backward that ends a level's chain).  One workgroup.
// QC_EPI_SUMB (R <= kQcRB): y[c][n] = sum over the samples b of xs[b nc + c][n] — the gradient of a query that was broadcast over the batch.
__global__ __launch_bounds__(256) void qc_stage_rows_kernel(QcStage s) {
  ICL_DYN_LDS(float, xs);
  const int K = s.K;
  for (int r0 = 0; r0 < s.R; r0 += kQcRB) {
    const int nr = s.R - r0 < kQcRB ? s.R - r0 : kQcRB;
    if (r0) __syncthreads();
    qc_prologue(s, xs, r0, nr);
    if (s.epi == QC_EPI_SUMB) {
      for (int i = threadIdx.x; i < s.nc * K; i += 256) {
        const int c = i / K, n = i - c * K;
        float v = 0.f;
        for (int b = 0; b * s.nc + c < s.R; ++b) v += xs[(b * s.nc + c) * K + n];
        s.y[i] = v;
      }
    } else {
      for (int i = threadIdx.x; i < nr * K; i += 256) {
        const int row = i / K, n = i - row * K;
        s.y[(long)(r0 + row) * K + n] = qc_epilogue(s, xs[i], r0 + row, n);
      }
    }
  }
}

// ---------------------------------------------------------------- every parameter gradient of a level in one launch
// kind 0: dw[n][k] = sum_r g[r][n] x[r][k] (+ db[n] = sum_r g[r][n]) — a Linear's weight / bias gradient from its <= 32 rows;
// kind 1: dw[k] = sum_r g[r][k] x[r][k], db[k] = sum_r g[r][k] — LayerNorm's gamma / beta from (d(normalised rows), xhat).
// Sums over r in row order.  grid (blocks, jobs).
constexpr int kQcMaxJobs = 12;
struct QcWgradJobs {
  const float* g[kQcMaxJobs]; const float* x[kQcMaxJobs];
  float* dw[kQcMaxJobs]; float* db[kQcMaxJobs];
  int R[kQcMaxJobs], N[kQcMaxJobs], K[kQcMaxJobs], kind[kQcMaxJobs];
};
__global__ __launch_bounds__(256) void qc_wgrad_kernel(QcWgradJobs j) {
  const int e = blockIdx.y;
  const float* __restrict__ g = j.g[e];
  const float* __restrict__ x = j.x[e];
  float* __restrict__ dw = j.dw[e];
  float* __restrict__ db = j.db[e];
  const int R = j.R[e], N = j.N[e], K = j.K[e];
  if (j.kind[e] == 1) {
    for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < R; ++r) { const float gv = g[(long)r * K + k]; a += gv * x[(long)r * K + k]; b += gv; }
      dw[k] = a;
      if (db) db[k] = b;
    }
    return;
  }
  const int kq = K >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)N * kq; i += (long)gridDim.x * 256) {
    const int n = (int)(i / kq), k = 4 * (int)(i - (long)n * kq);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float b = 0.f;
    for (int r = 0; r < R; ++r) {
      const float gv = g[(long)r * N + n];
      const float4 xv = *reinterpret_cast<const float4*>(x + (long)r * K + k);
      a.x += gv * xv.x; a.y += gv * xv.y; a.z += gv * xv.z; a.w += gv * xv.w;
      b += gv;
    }
    *reinterpret_cast<float4*>(dw + (long)n * K + k) = a;
    if (db && k == 0) db[n] = b;
  }
}

}  // namespace icl
