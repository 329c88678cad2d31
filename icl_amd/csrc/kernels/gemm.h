// gemm.h — the dense fp32 products of the ICL hot path on v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain).
//
// Reference operators replaced (paths relative to /root/reference/code):
//   nn.Linear forward / input gradient / weight gradient of the aligner heads — fc_q, fc_kv, proj, MLP.fc1/fc2 and the token-axis
//   Class_Decoder.mlp2 (networks/unet_3D_icl.py:258-268,283-315), query_convs (Conv1d k=1, :197), the token projection (:212);
//   the Swin qkv / proj / MLP / PatchMerging / PatchEmbed linears (networks/swinunetr_icl.py:703-750,812,946-976,1170);
//   ConvTranspose3d(k=2, s=2) of MONAI UnetrUpBlock as one product (swinunetr_icl.py:178-225); the 3^3 convolutions on <= 6^3
//   voxels (`center`, networks/utils.py:104,107) through im2col.
//
// Two kernels:
//   linear_stream_kernel     y = x W^T and gx = gy W for <= 32 rows: the weight matrix is STREAMED once from HBM (4 rows x 256
//                            contiguous bytes per wave instruction), the small operand stays resident in LDS;
//   gemm_kernel              C = A B for everything else: LDS-staged 64x64 wave tiles, operands either "k-contiguous" (row-major
//                            [rows][K]) or "k-strided" ([K][rows]); optional split-K into slabs summed in a fixed order.
// The MFMA k index is only a label, so a lane's float4 (16-byte) load supplies four k values (k-contiguous operand) or four
// output columns / rows (k-strided operand) to four MFMAs and no register transposes are needed.
#pragma once

namespace icl {

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

// epilogue shared by all product kernels: bias (per output column) and activation (0 none, 1 exact-erf GELU)
__device__ __forceinline__ float gemm_epilogue(float v, const float* __restrict__ bias, int col, int act) {
  if (bias) v += bias[col];
  if (act == 1) v = gelu_erf(v);
  return v;
}

// out[m][n..n+3] = act(bias + sum_s slab[s][m][n..n+3]): the slabs are added in slab order, one output quad per call.
__device__ __forceinline__ void reduce_slabs_quad(const float* slab, float* out, const float* __restrict__ bias, int S, long m,
                                                  int n, long spitch, long sstride, long ldo, int act_flags) {
  const int act = act_flags & 15;
  const bool brow = (act_flags & 16) != 0;            // bias indexed by the output row instead of the column
  float4 a = *reinterpret_cast<const float4*>(slab + m * spitch + n);
  int s = 1;
  // eight slabs' loads in flight, added in slab order (a loop of one dependent load per slab took 15 us for the 9-18 slices of a
  // 13,824-wide product)
  for (; s + 8 <= S; s += 8) {
    float4 b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(slab + (s + u) * sstride + m * spitch + n);
#pragma unroll
    for (int u = 0; u < 8; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
  }
  for (; s < S; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(slab + s * sstride + m * spitch + n);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  a.x = gemm_epilogue(a.x, bias, brow ? (int)m : n, act);
  a.y = gemm_epilogue(a.y, bias, brow ? (int)m : n + 1, act);
  a.z = gemm_epilogue(a.z, bias, brow ? (int)m : n + 2, act);
  a.w = gemm_epilogue(a.w, bias, brow ? (int)m : n + 3, act);
  *reinterpret_cast<float4*>(out + m * ldo + n) = a;
}

// ------------------------------------------------------------------------------------------------ weight streaming
// Skinny products against a big row-major weight W [O][K] (the 13,824^2 token-axis MLP, <= 32 activation rows):
//   NN = false  forward         part[s][m][o] = sum_{k in slice s} xs[m][k] W[o][k]      xs = x  [M][K]
//   NN = true   input gradient  part[s][m][k] = sum_{o in slice s} xs[m][o] W[o][k]      xs = gy [M][O]
// W is read exactly once, at the HBM rate.  Measured on MI355X (tools/probe/rowwidth_probe.hip): a wave instruction must cover
// >= 256 contiguous bytes per weight row to stream at 6.6 TB/s (128 B: 6.0, the 64 B of an MFMA-fragment-shaped load: 3.6-4.6),
// so every load instruction here is 4 rows x 256 B (lane l: row l >> 4, 16-byte quad l & 15) and the 16 x 64 tile goes through a
// wave-private 4 KiB LDS buffer into the MFMA operand layout (LDS operations of one wave execute in order: no barrier in the loop).
// The contraction axis is cut into `nslice` slices; the workgroup (8 waves) keeps the slice of the small operand xs resident in
// LDS (k-contiguous rows, quads XOR-swizzled with the row so a fragment read is conflict free) and each wave walks `upw` units
// of the other weight axis: a unit is 16 weight rows (forward: one MFMA tile of outputs, the wave walks the slice's 64-column
// tiles) or 64 weight columns (input gradient: four MFMA tiles, the wave walks down the slice's 16-row tiles).  P tiles are in
// flight per wave (registers).  Partials go to slab[slice]; gemm_reduce_slabs_kernel adds them in a fixed order (+ bias, act).
// grid (ceil(units / (8 upw)), nslice), block 512, dynamic LDS = (16 MT slice_len + 8 * 1024) floats; slice_len % 64 == 0, K % 4 == 0.
// With a single slice (small weights) `slab` is the final output [M][.]: rows >= M are not stored and bias / act are applied here.
template <int MT, bool NN, int P>
__global__ __launch_bounds__(512) void linear_stream_kernel(const float* __restrict__ xs, const float* __restrict__ w, float* __restrict__ slab,
                                                            int M, int K, int O, int slice_len, int upw, const float* __restrict__ bias,
                                                            int act,
                                                            const float* __restrict__ wg_x, float* __restrict__ wg_dw, int wg_blocks) {
  ICL_DYN_LDS(float, lds);
  float* xl = lds;                                        // [16 MT][slice_len], quad q of row m stored at quad q ^ (m & 15)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, u = lane & 15, lg = lane >> 4;
  if (NN && wg_dw != nullptr && (int)blockIdx.x >= (int)gridDim.x - wg_blocks) {
    // The last `wg_blocks` workgroup columns of an input-gradient launch compute the layer's WEIGHT gradient dw[o][k] = sum_r xs[r][o] wg_x[r][k]
    // (xs = dY [M][O], wg_x = the layer input [M][K]): the arithmetic of linear_wgrad_outer_kernel, in the same launch — one dependent
    // launch fewer per small Linear layer on the aligner heads' backward chains (profiles/r4_timeline.md).
    if (blockIdx.y != 0) return;
    const int wb = (int)blockIdx.x - ((int)gridDim.x - wg_blocks);
    const int iq = K >> 2;
    const long total = (long)O * iq;
    for (long e = (long)wb * 512 + tid; e < total; e += (long)wg_blocks * 512) {
      const int o = (int)(e / iq), i = (int)(e % iq) * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int r = 0; r < M; ++r) {
        const float g = xs[(long)r * O + o];
        const float4 xv = *reinterpret_cast<const float4*>(wg_x + (long)r * K + i);
        acc.x = fmaf(g, xv.x, acc.x);
        acc.y = fmaf(g, xv.y, acc.y);
        acc.z = fmaf(g, xv.z, acc.z);
        acc.w = fmaf(g, xv.w, acc.w);
      }
      *reinterpret_cast<float4*>(wg_dw + (long)o * K + i) = acc;
    }
    return;
  }
  float* wl = lds + 16 * MT * slice_len + wid * 1024;     // this wave's 16 x 64 weight tile
  const int cdim = NN ? O : K;                            // contraction axis (columns of xs)
  const int c0 = blockIdx.y * slice_len;
  const int clen = c0 + slice_len < cdim ? slice_len : cdim - c0;   // > 0 by construction of the grid
  const int tpu = NN ? (clen + 15) / 16 : (clen + 63) / 64;         // tiles per unit
  const int nunits = NN ? (K + 63) / 64 : (O + 15) / 16;
  const int unit0 = (blockIdx.x * 8 + wid) * upw;
  int my_units = nunits - unit0;
  if (my_units > upw) my_units = upw;
  const int total = my_units > 0 ? my_units * tpu : 0;

  // ---- weight tile stream (loads are issued first so that they overlap the staging of xs)
  float4 ring[P][4];
  int l_unit = unit0, l_t = 0;                            // position of the next tile to load
  auto issue = [&](int p, bool live) {
    // tile rows/cols: forward: rows 16 unit + 4 j + lg, cols c0 + 64 t + 4 u;  gradient: rows c0 + 16 t + 4 j + lg, cols 64 unit + 4 u
    const int row0 = NN ? c0 + 16 * l_t : 16 * l_unit;
    const int col = (NN ? 64 * l_unit : c0 + 64 * l_t) + 4 * u;
    const int row_end = NN ? c0 + clen : O, col_end = NN ? K : c0 + clen;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = row0 + 4 * j + lg;
      const bool ok = live && row < row_end && col < col_end;
      ring[p][j] = icl_nt_load4(w + (ok ? (long)row * K + col : 0L));
      if (!ok) ring[p][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (++l_t == tpu) { l_t = 0; ++l_unit; }
  };
#pragma unroll
  for (int p = 0; p < P; ++p) issue(p, p < total);

  // ---- resident slice of the small operand
  const int qpr = slice_len >> 2;                         // quads per row
  for (int it = tid; it < 16 * MT * qpr; it += 512) {
    const int m = it / qpr, q = it % qpr;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M && 4 * q < clen) v = *reinterpret_cast<const float4*>(xs + (long)m * cdim + c0 + 4 * q);
    *reinterpret_cast<float4*>(xl + m * slice_len + 4 * (q ^ (m & 15))) = v;
  }
  __syncthreads();

  f32x4 acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  int c_unit = unit0, c_t = 0;                            // position of the tile being computed
  const bool direct = gridDim.y == 1;                      // one slice: write the result itself
  float* out = slab + (long)blockIdx.y * 16 * MT * (NN ? K : O);

  for (int base = 0; base < total; base += P) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (base + p >= total) break;
      // registers -> wave-private LDS tile [16][64]; forward swizzles the quad with the row (fragment rows = lanes u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 4 * j + lg;
        *reinterpret_cast<float4*>(wl + row * 64 + 4 * (NN ? u : (u ^ row))) = ring[p][j];
      }
      ICL_WAVE_SYNC();
      issue(p, base + p + P < total);
      if (!NN) {
        // 64 columns of the tile = 4 groups of 16 k labels: lane (u, lg) takes k = 64 t + 16 s + 4 lg + e of weight row u / xs row u
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float4 b = *reinterpret_cast<const float4*>(wl + u * 64 + 4 * ((4 * s + lg) ^ u));
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const int m = 16 * t + u;
            const float4 a = *reinterpret_cast<const float4*>(xl + m * slice_len + 4 * ((16 * c_t + 4 * s + lg) ^ u));
            acc[t][0] = icl_mfma_16x16x4(a.x, b.x, acc[t][0]);
            acc[t][1] = icl_mfma_16x16x4(a.y, b.y, acc[t][1]);
            acc[t][2] = icl_mfma_16x16x4(a.z, b.z, acc[t][2]);
            acc[t][3] = icl_mfma_16x16x4(a.w, b.w, acc[t][3]);
          }
        }
      } else {
        // 16 weight rows of the tile = k labels 16 t + 4 lg + e; each row read gives four consecutive output columns 4 u + e'
        float4 b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = *reinterpret_cast<const float4*>(wl + (4 * lg + e) * 64 + 4 * u);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int m = 16 * t + u;
          const float4 a = *reinterpret_cast<const float4*>(xl + m * slice_len + 4 * ((4 * c_t + lg) ^ u));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float av = f4c(a, e);
            acc[t][0] = icl_mfma_16x16x4(av, b[e].x, acc[t][0]);
            acc[t][1] = icl_mfma_16x16x4(av, b[e].y, acc[t][1]);
            acc[t][2] = icl_mfma_16x16x4(av, b[e].z, acc[t][2]);
            acc[t][3] = icl_mfma_16x16x4(av, b[e].w, acc[t][3]);
          }
        }
      }
      if (++c_t == tpu) {
        // unit done: D register q of lane (u, lg) is row 16 t + 4 lg + q
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m = 16 * t + 4 * lg + q;
            if (direct && m >= M) continue;
            if (NN) {
              const int col = 64 * c_unit + 4 * u;
              float4 v = make_float4(acc[t][0][q], acc[t][1][q], acc[t][2][q], acc[t][3][q]);
              if (direct) {
                v.x = gemm_epilogue(v.x, bias, col, act);
                v.y = gemm_epilogue(v.y, bias, col + 1, act);
                v.z = gemm_epilogue(v.z, bias, col + 2, act);
                v.w = gemm_epilogue(v.w, bias, col + 3, act);
              }
              if (col < K) *reinterpret_cast<float4*>(out + (long)m * K + col) = v;
            } else {
              const int o = 16 * c_unit + u;
              if (o < O) {
                const float v = (acc[t][0][q] + acc[t][1][q]) + (acc[t][2][q] + acc[t][3][q]);
                out[(long)m * O + o] = direct ? gemm_epilogue(v, bias, o, act) : v;
              }
            }
          }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        c_t = 0;
        ++c_unit;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ input gradient + SGD update, one pass
// The backward of a big skinny Linear and its optimiser step read the same 764 MB matrix twice (gx = gy W, then p / m in and
// out of the update).  On one rank the factors of dW = gy^T x are complete the moment the layer's backward runs, so this kernel
// does both while the weight tile is on chip: every 16 x 64 tile of p (and m) is loaded once, its contribution to gx is
// accumulated from the OLD values (slab[s], summed by gemm_reduce_slabs_kernel as in linear_stream_kernel), then
//     d = gy^T x (MFMA, 16 factor rows per step) + wd * p;   m = first ? d : momentum * m + d;   p -= lr * m
// (torch.optim.SGD, optim.h) is applied to the tile and p / m go back with non-temporal stores: 16 B per weight of HBM traffic for
// the layer's backward + update instead of 20.  Same grid / slices / units as linear_stream_kernel<MT, true>; x is [M][K] (the
// layer input), gy [M][O]; each weight element belongs to exactly one wave.  Only legal when the weight is used once per step.
template <int MT, int P>
__global__ __launch_bounds__(512) void linear_dgrad_sgd_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ w,
                                                               float* __restrict__ mom, float* __restrict__ slab, int M, int K, int O,
                                                               int slice_len, int upw, float lr, float momentum, float wd, int first,
                                                               const float* __restrict__ lr_dev) {
  ICL_DYN_LDS(float, lds);
  float* xl = lds;                                        // gy slice [16 MT][slice_len], quad q of row m stored at quad q ^ (m & 15)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, u = lane & 15, lg = lane >> 4;
  float* wl = lds + 16 * MT * slice_len + wid * 1024;     // this wave's 16 x 64 tile buffer
  if (lr_dev) lr = *lr_dev;
  const int c0 = blockIdx.y * slice_len;
  const int clen = c0 + slice_len < O ? slice_len : O - c0;
  const int tpu = (clen + 15) / 16;
  const int nunits = (K + 63) / 64;
  const int unit0 = (blockIdx.x * 8 + wid) * upw;
  int my_units = nunits - unit0;
  if (my_units > upw) my_units = upw;
  const int total = my_units > 0 ? my_units * tpu : 0;

  float4 ring[P][4], mring[P][4];
  int l_unit = unit0, l_t = 0;
  auto issue = [&](int p, bool live) {
    const int row0 = c0 + 16 * l_t, col = 64 * l_unit + 4 * u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = row0 + 4 * j + lg;
      const bool ok = live && row < c0 + clen && col < K;
      const long off = ok ? (long)row * K + col : 0L;
      // straight-line loads from an always-valid address, then a select (as in linear_stream_kernel): out-of-range lanes hold
      // zeros — gx accumulates gy * ring, and 0 * w[0] is only zero while w[0] is finite
      ring[p][j] = icl_nt_load4(w + off);
      mring[p][j] = first ? make_float4(0.f, 0.f, 0.f, 0.f) : icl_nt_load4(mom + off);
      if (!ok) ring[p][j] = mring[p][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (++l_t == tpu) { l_t = 0; ++l_unit; }
  };
#pragma unroll
  for (int p = 0; p < P; ++p) issue(p, p < total);

  const int qpr = slice_len >> 2;
  for (int it = tid; it < 16 * MT * qpr; it += 512) {
    const int m = it / qpr, q = it % qpr;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M && 4 * q < clen) v = *reinterpret_cast<const float4*>(gy + (long)m * O + c0 + 4 * q);
    *reinterpret_cast<float4*>(xl + m * slice_len + 4 * (q ^ (m & 15))) = v;
  }
  __syncthreads();

  f32x4 acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  int c_unit = unit0, c_t = 0;
  float* out = slab + (long)blockIdx.y * 16 * MT * K;
  // x rows of the current 64-column strip as MFMA B operands of d = gy^T x: xq[s][c] = x[4 s + lg][64 unit + 16 c + u]
  float xq[4 * MT][4];
  auto load_x = [&](int unit) {
#pragma unroll
    for (int s = 0; s < 4 * MT; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int m = 4 * s + lg, col = 64 * unit + 16 * c + u;
        xq[s][c] = (m < M && col < K) ? x[(long)m * K + col] : 0.f;
      }
  };
  if (total > 0) load_x(c_unit);

  for (int base = 0; base < total; base += P) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (base + p >= total) break;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(wl + (4 * j + lg) * 64 + 4 * u) = ring[p][j];
      ICL_WAVE_SYNC();
      // gx += gy[:, rows of the tile] * W_old[tile]
      {
        float4 b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = *reinterpret_cast<const float4*>(wl + (4 * lg + e) * 64 + 4 * u);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const float4 a = *reinterpret_cast<const float4*>(xl + (16 * t + u) * slice_len + 4 * ((4 * c_t + lg) ^ u));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float av = f4c(a, e);
            acc[t][0] = icl_mfma_16x16x4(av, b[e].x, acc[t][0]);
            acc[t][1] = icl_mfma_16x16x4(av, b[e].y, acc[t][1]);
            acc[t][2] = icl_mfma_16x16x4(av, b[e].z, acc[t][2]);
            acc[t][3] = icl_mfma_16x16x4(av, b[e].w, acc[t][3]);
          }
        }
      }
      // d tile = gy^T x: A[o = u][factor row 4 s + lg] from the resident gy slice, B = xq; D[o = 4 lg + q][col 16 c + u]
      f32x4 dacc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) dacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4 * MT; ++s) {
        const int m = 4 * s + lg;
        const float av = xl[m * slice_len + 4 * ((4 * c_t + (u >> 2)) ^ (m & 15)) + (u & 3)];
#pragma unroll
        for (int c = 0; c < 4; ++c) dacc[c] = icl_mfma_16x16x4(av, xq[s][c], dacc[c]);
      }
      ICL_WAVE_SYNC();                                     // every lane has read its W fragments: the tile buffer is free
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) wl[(4 * lg + q) * 64 + 16 * c + u] = dacc[c][q];
      ICL_WAVE_SYNC();
      {
        const int row0 = c0 + 16 * c_t, col = 64 * c_unit + 4 * u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = row0 + 4 * j + lg;
          if (row < c0 + clen && col < K) {
            const float4 d = *reinterpret_cast<const float4*>(wl + (4 * j + lg) * 64 + 4 * u);
            float4 pv = ring[p][j], mv = mring[p][j];
            sgd_update4(pv, d, mv, lr, momentum, wd, first);
            const long off = (long)row * K + col;
            icl_nt_store4(w + off, pv);
            icl_nt_store4(mom + off, mv);
          }
        }
      }
      ICL_WAVE_SYNC();                                     // the d tile has been read: the next tile may overwrite the buffer
      issue(p, base + p + P < total);
      if (++c_t == tpu) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m = 16 * t + 4 * lg + q, col = 64 * c_unit + 4 * u;
            if (col < K) *reinterpret_cast<float4*>(out + (long)m * K + col) = make_float4(acc[t][0][q], acc[t][1][q], acc[t][2][q], acc[t][3][q]);
          }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        c_t = 0;
        ++c_unit;
        if (base + p + 1 < total) load_x(c_unit);
      }
    }
  }
}

// out[m][n] = act(bias[n] + sum_s slab[s][m][n]) for m < M: fixed-order sum of split partials (bitwise reproducible).
// slab rows have pitch `spitch` and slabs are `sstride` floats apart; out rows have pitch ldo.  N % 4 == 0.
__global__ __launch_bounds__(256) void gemm_reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ out, const float* __restrict__ bias,
                                                                int S, long M, int N, long spitch, long sstride, long ldo, int act_flags) {
  const int nq = N >> 2;
  const long total = M * nq;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x)
    reduce_slabs_quad(slab, out, bias, S, it / nq, (int)(it % nq) * 4, spitch, sstride, ldo, act_flags);
}

// ------------------------------------------------------------------------------------------------ weight gradient, few rows
// dw[o][i] = sum_{r < rows} gy[r][o] * x[r][i] for a handful of rows (the aligner's Linear layers see batch*classes = 4..32 rows):
// a sum of `rows` outer products, one thread per (o, four consecutive i), 16-byte loads / stores.  I % 4 == 0.
__global__ __launch_bounds__(256) void linear_wgrad_outer_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ dw,
                                                                 int rows, int O, int I) {
  const int iq = I >> 2;
  const long total = (long)O * iq;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int o = (int)(e / iq), i = (int)(e % iq) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < rows; ++r) {
      const float g = gy[(long)r * O + o];
      const float4 xv = *reinterpret_cast<const float4*>(x + (long)r * I + i);
      acc.x = fmaf(g, xv.x, acc.x);
      acc.y = fmaf(g, xv.y, acc.y);
      acc.z = fmaf(g, xv.z, acc.z);
      acc.w = fmaf(g, xv.w, acc.w);
    }
    *reinterpret_cast<float4*>(dw + (long)o * I + i) = acc;
  }
}

// ------------------------------------------------------------------------------------------------ general product
// C[m][n] = sum_k A(m,k) B(k,n), block tile (WM*64) x (WN*64), wave tile 64 x 64 (4 x 4 MFMA tiles), k-steps of 32 staged in LDS.
//   AK  = true : A is row-major [M][lda], k contiguous.   LDS image rows x 32 floats, the eight 16-byte quads of a row XOR-swizzled
//                with (row >> 1) & 7: the fragment read (lane (u = l & 15, lg): row 16 i + u, quad 4 s + lg, one ds_read_b128 = four
//                k labels) touches 16 different 16-byte slots per lane group — conflict free;
//   AK  = false: A is given transposed, [K][lda] with m contiguous.  LDS image 32 k-rows x (WM*64) floats; the fragment read of
//                k row 16 s + 4 lg + e takes four consecutive m (tile c of the 64-row group holds rows 4 u + c): conflict free.
//   BK_ likewise for B: true = [N][ldb] (k contiguous: y = x W^T), false = [K][ldb] (n contiguous: gx = gy W).
// Split-K: split z of nsplit takes the z-th k range of `kper` (multiple of 32) and writes a raw partial tile to C + z * c_split_stride.
// Epilogue (only when not split): bias per column (act flag 16: per row) + activation (low bits of act).  Global tiles are fetched into registers one k-step ahead.
struct GemmArgs {
  const float* a;
  const float* b;
  float* c;
  const float* bias;
  long lda, ldb, ldc, c_split_stride;
  long a_bstride, b_bstride, c_bstride;   // batched products: blockIdx.z = batch * nsplit + split
  int M, N, K, kper, nsplit, act;
};

__device__ __forceinline__ float4 gemm_load4(const float* __restrict__ base, long ld, int row, int col, int rows, int cols, bool vec_ok) {
  // element (row, col..col+3) of a row-major [rows][ld] matrix; zeros outside
  if (row >= rows || col >= cols) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float* p = base + (long)row * ld + col;
  if (vec_ok && col + 3 < cols) return *reinterpret_cast<const float4*>(p);
  float4 v;
  v.x = p[0];
  v.y = col + 1 < cols ? p[1] : 0.f;
  v.z = col + 2 < cols ? p[2] : 0.f;
  v.w = col + 3 < cols ? p[3] : 0.f;
  return v;
}

// swizzled quad of a k-contiguous LDS row: rows of 8 quads (KT = 32) or 16 quads (KT = 64)
template <int KT>
__device__ __forceinline__ int gemm_swz(int row, int q) { return KT == 32 ? (q ^ ((row >> 1) & 7)) : (q ^ (row & 15)); }

// FAST (chosen by the launcher): both operands 16-byte aligned with pitches % 4 == 0, K % 4 == 0 for a k-contiguous operand and
// the row count % 4 == 0 for a k-strided one.  The k-loop then holds nothing but 16-byte loads at loop-invariant pointers + k:
// row indices are CLAMPED (rows past M / N only feed output rows / columns that are never stored) and only the last, partial
// k-step zeroes the quads past the end of the range.  !FAST: element-wise bounds-checked loads (gemm_load4).
// TN = MFMA tiles per wave along n (4; 3 for k-contiguous B: 48-column wave tiles fit the 48 / 96 / 144 / 192 / 288-wide Linear layers
// of the Swin stages without padded columns).
template <bool AK, bool BK_, int WM, int WN, bool FAST = true, int KT = 32, int TN = 4>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmArgs g) {
  static_assert(TN == 4 || BK_, "a k-strided B operand supplies four n tiles per 16-byte read");
  constexpr int NT = WM * WN * 64, BM = WM * 64, WNC = 16 * TN, BN = WN * WNC, QR = KT / 4;   // QR quads per k-contiguous row
  constexpr int AV = (BM * KT / 4 + NT - 1) / NT, BV = (BN * KT / 4 + NT - 1) / NT;       // float4 per thread and k-step
  constexpr bool ARAG = (BM * KT / 4) % NT != 0, BRAG = (BN * KT / 4) % NT != 0;   // the last item of a thread may lie past the tile
  ICL_DYN_LDS(float, lds);
  float* as = lds;                 // AK: [BM][32] swizzled; else [32][BM]
  float* bs = lds + BM * KT;       // BK_: [BN][32] swizzled; else [32][BN]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, u = lane & 15, lg = lane >> 4;
  const int wm = wid / WN, wn = wid % WN;
  int m0 = blockIdx.y * BM;            // row block being computed; the workgroup walks row blocks m0, m0 + gridDim.y * BM, ...
  int mf = m0;                         // row block the staging loads address (one block ahead at the end of a k range)
  const int n0 = blockIdx.x * BN;
  const int batch = blockIdx.z / g.nsplit, split_id = blockIdx.z % g.nsplit;
  const int k_lo = split_id * g.kper;
  const int k_hi = k_lo + g.kper < g.K ? k_lo + g.kper : g.K;
  g.a += batch * g.a_bstride;
  g.b += batch * g.b_bstride;
  float4 ar[AV], br[BV];
  // per-thread staging pointers (loop invariant)
  const float* ap[AV];
  const float* bp[BV];
  auto item_ptr = [&](const float* base, long ld, bool kc, int r0, int rows, int it, int BX) -> const float* {
    if (kc) {   // item = (row, k quad): QR quads per row
      int row = r0 + it / QR;
      row = row < rows ? row : rows - 1;
      return base + (long)row * ld + 4 * (it % QR);
    }
    // item = (k row, quad of the row index): BX/4 quads per k row; clamped quads feed rows that are never stored
    int c = r0 + 4 * (it % (BX / 4));
    c = c + 3 < rows ? c : rows - 4;
    return base + (long)(it / (BX / 4)) * ld + c;
  };
  auto set_a_ptrs = [&]() {
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      const int it = tid + j * NT;
      ap[j] = item_ptr(g.a, g.lda, AK, mf, g.M, (ARAG && it >= BM * QR) ? 0 : it, BM);
    }
  };
  if (FAST) {
    set_a_ptrs();
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      const int it = tid + j * NT;
      bp[j] = item_ptr(g.b, g.ldb, BK_, n0, g.N, (BRAG && it >= BN * QR) ? 0 : it, BN);
    }
  }
  // k offset of an item inside a k-step (for the tail test) and the pointer stride per k
  auto item_k = [&](bool kc, int it, int BX) { return kc ? 4 * (it % QR) : it / (BX / 4); };
  const long astep = AK ? 1 : g.lda, bstep = BK_ ? 1 : g.ldb;
  auto fetch = [&](int k0) {
    if (FAST) {
      if (k0 + KT <= k_hi) {                              // every k-step but possibly the last one
#pragma unroll
        for (int j = 0; j < AV; ++j) ar[j] = *reinterpret_cast<const float4*>(ap[j] + k0 * astep);
#pragma unroll
        for (int j = 0; j < BV; ++j) br[j] = *reinterpret_cast<const float4*>(bp[j] + k0 * bstep);
      } else {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
          const bool ok = k0 + item_k(AK, tid + j * NT, BM) < k_hi;
          const float4 v = *reinterpret_cast<const float4*>(ok ? ap[j] + k0 * astep : g.a);
          ar[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
          const bool ok = k0 + item_k(BK_, tid + j * NT, BN) < k_hi;
          const float4 v = *reinterpret_cast<const float4*>(ok ? bp[j] + k0 * bstep : g.b);
          br[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < AV; ++j) {
        const int it = tid + j * NT;
        if (AK) ar[j] = gemm_load4(g.a, g.lda, mf + it / QR, k0 + 4 * (it % QR), g.M, k_hi, false);
        else ar[j] = gemm_load4(g.a, g.lda, k0 + it / (BM / 4), mf + 4 * (it % (BM / 4)), k_hi, g.M, false);
      }
#pragma unroll
      for (int j = 0; j < BV; ++j) {
        const int it = tid + j * NT;
        if (BK_) br[j] = gemm_load4(g.b, g.ldb, n0 + it / QR, k0 + 4 * (it % QR), g.N, k_hi, false);
        else br[j] = gemm_load4(g.b, g.ldb, k0 + it / (BN / 4), n0 + 4 * (it % (BN / 4)), k_hi, g.N, false);
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      const int it = tid + j * NT;
      if (ARAG && it >= BM * QR) continue;
      if (AK) {
        const int row = it / QR, q = it % QR;
        *reinterpret_cast<float4*>(as + row * KT + 4 * gemm_swz<KT>(row, q)) = ar[j];
      } else {
        *reinterpret_cast<float4*>(as + 4 * it) = ar[j];
      }
    }
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      const int it = tid + j * NT;
      if (BRAG && it >= BN * QR) continue;
      if (BK_) {
        const int row = it / QR, q = it % QR;
        *reinterpret_cast<float4*>(bs + row * KT + 4 * gemm_swz<KT>(row, q)) = br[j];
      } else {
        *reinterpret_cast<float4*>(bs + 4 * it) = br[j];
      }
    }
  };
  f32x4 acc[4][TN];
  fetch(k_lo);
  // Row blocks are walked inside the workgroup (tall products launch ~2 workgroups per CU): the first staging loads of the NEXT
  // row block are issued before the epilogue of the current one, so a block's load latency hides behind the previous block's
  // stores instead of opening every (short-lived) workgroup.
  for (;;) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = k_lo; k0 < k_hi; k0 += KT) {
    __syncthreads();
    stash();
    __syncthreads();
    if (k0 + KT < k_hi) {
      fetch(k0 + KT);
    } else if (m0 + (int)gridDim.y * BM < g.M) {
      mf = m0 + (int)gridDim.y * BM;
      if (FAST) set_a_ptrs();
      fetch(k_lo);
    }
#pragma unroll
    for (int s = 0; s < KT / 16; ++s) {
      // operand fragments of this 16-deep k group: af[e][i] / bf[e][t] = value of MFMA tile i / t for k label 16 s + 4 lg + e
      float af[4][4], bf[4][TN];
      if (AK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wm * 64 + 16 * i + u;
          const float4 v = *reinterpret_cast<const float4*>(as + row * KT + 4 * gemm_swz<KT>(row, 4 * s + lg));
          af[0][i] = v.x; af[1][i] = v.y; af[2][i] = v.z; af[3][i] = v.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float4 v = *reinterpret_cast<const float4*>(as + (16 * s + 4 * lg + e) * BM + wm * 64 + 4 * u);
          af[e][0] = v.x; af[e][1] = v.y; af[e][2] = v.z; af[e][3] = v.w;
        }
      }
      if constexpr (BK_) {
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int row = wn * WNC + 16 * t + u;
          const float4 v = *reinterpret_cast<const float4*>(bs + row * KT + 4 * gemm_swz<KT>(row, 4 * s + lg));
          bf[0][t] = v.x; bf[1][t] = v.y; bf[2][t] = v.z; bf[3][t] = v.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float4 v = *reinterpret_cast<const float4*>(bs + (16 * s + 4 * lg + e) * BN + wn * WNC + 4 * u);
          bf[e][0] = v.x; bf[e][1] = v.y; bf[e][2] = v.z; bf[e][3] = v.w;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < TN; ++t) acc[i][t] = icl_mfma_16x16x4(af[e][i], bf[e][t], acc[i][t]);
    }
  }
  // D register q of tile (i, t), lane (u, lg): A-side index 4 lg + q, B-side index u.
  //   AK : row = 16 i + (4 lg + q)       else: row = 4 (4 lg + q) + i
  //   BK_: col = 16 t + u                else: col = 4 u + t   (t = 0..3 consecutive: 16-byte stores)
  float* c = g.c + batch * g.c_bstride + (long)split_id * g.c_split_stride;
  const int act = g.act & 15;
  const bool brow = (g.act & 16) != 0;
  const bool split = g.c_split_stride != 0 || g.nsplit > 1;
  if constexpr (BK_) {
    // k-contiguous B: a lane holds ONE column per tile (16 lanes = 64 contiguous bytes per row and store instruction, the access
    // shape that streams at half the HBM rate).  The wave's 64 x WNC tile goes through a wave-private LDS buffer in two 32-row
    // halves and leaves as 16-byte stores along the rows: 192 / 256 contiguous bytes per row and instruction.
    constexpr int EP = WNC + 4, QW = WNC / 4;
    const bool cvec = (g.ldc & 3) == 0 && (((unsigned long long)c) & 15ull) == 0;
    __syncthreads();                                   // every wave is done with the operand tiles
    float* ep = lds + wid * (32 * EP);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int lrow = AK ? 16 * i + 4 * lg + q : 4 * (4 * lg + q) + i;      // row inside the wave tile
          if ((lrow >> 5) != h) continue;
#pragma unroll
          for (int t = 0; t < TN; ++t) ep[(lrow & 31) * EP + 16 * t + u] = acc[i][t][q];
        }
      ICL_WAVE_SYNC();
      for (int it = lane; it < 32 * QW; it += 64) {
        const int r = it / QW, qd = it % QW;
        const int row = m0 + wm * 64 + 32 * h + r, col = n0 + wn * WNC + 4 * qd;
        if (row >= g.M || col >= g.N) continue;
        float4 v = *reinterpret_cast<const float4*>(ep + r * EP + 4 * qd);
        if (!split) {
          v.x = gemm_epilogue(v.x, g.bias, brow ? row : col, act);
          v.y = gemm_epilogue(v.y, col + 1 < g.N ? g.bias : nullptr, brow ? row : col + 1, act);
          v.z = gemm_epilogue(v.z, col + 2 < g.N ? g.bias : nullptr, brow ? row : col + 2, act);
          v.w = gemm_epilogue(v.w, col + 3 < g.N ? g.bias : nullptr, brow ? row : col + 3, act);
        }
        float* dst = c + (long)row * g.ldc + col;
        if (cvec && col + 3 < g.N) {
          *reinterpret_cast<float4*>(dst) = v;
        } else {
          dst[0] = v.x;
          if (col + 1 < g.N) dst[1] = v.y;
          if (col + 2 < g.N) dst[2] = v.z;
          if (col + 3 < g.N) dst[3] = v.w;
        }
      }
      ICL_WAVE_SYNC();
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = m0 + wm * 64 + (AK ? 16 * i + 4 * lg + q : 4 * (4 * lg + q) + i);
        if (row >= g.M) continue;
        float* crow = c + (long)row * g.ldc;
        const int col = n0 + wn * WNC + 4 * u;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (split || col + t >= g.N) ? acc[i][t][q] : gemm_epilogue(acc[i][t][q], g.bias, brow ? row : col + t, act);
        if (col + 3 < g.N && (g.ldc & 3) == 0 && (((unsigned long long)c) & 15ull) == 0) {
          *reinterpret_cast<float4*>(crow + col) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (col + t < g.N) crow[col + t] = v[t];
        }
      }
  }
  m0 += (int)gridDim.y * BM;
  if (m0 >= g.M) break;
  }   // row blocks
}

// ------------------------------------------------------------------------------------------------ tall, skinny Linear: weights in registers
// y[rows][N] = act(x[rows][K] W^T + bias) for the token Linear layers of SwinUNETR's first stage: 221,184 / 235,298 rows, K, N in
// {48, 144, 192}: 42-170 MB of activations against 9-37 KB of weights.  The LDS-tiled product re-stages the weight tile for every
// 128-row block and keeps the matrix pipe 23 % busy (profiles/r2_pmc_gemm.md); here a wave keeps the WHOLE weight matrix as MFMA B
// fragments in registers (N K / 64 of them: 108-144) and streams 16-row blocks: K / 16 float4 loads of x per lane (the lane's four
// consecutive k feed four MFMAs: the k index of an MFMA is only a label that A and B must agree on), (N / 16) (K / 4) MFMAs, the
// 16 x N result transposed through a wave-private LDS tile and stored as whole rows.  No barrier, no weight traffic after the
// prologue.  KQ = K / 16, NT = N / 16.  WT: the weight is given as [K][N] (the input gradient g W of a weight stored [out][in]).
// NH: the N / 16 column tiles are dealt to NH waves (each re-reads the x rows, from L1 / L2): a third / half of the B fragments per
// wave, i.e. 60-100 instead of 170-220 registers on the 144- / 192-wide layers and twice the waves per SIMD.
template <int KQ, int NTALL, bool WT, int NH = 1>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, long rows, int act) {
  static_assert(NTALL % NH == 0, "column tiles must split evenly over the waves of a row block");
  constexpr int NT = NTALL / NH, K = 16 * KQ, NALL = 16 * NTALL, N = 16 * NT, LDP = N + 4;
  ICL_DYN_LDS(float, lds);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
  float* tl = lds + wid * 16 * LDP;                              // this wave's 16 x N output tile
  const long gw = (long)blockIdx.x * 4 + wid;                    // global wave: column part gw % NH of row-block stream gw / NH
  const int c0 = (int)(gw % NH) * N;
  w += WT ? c0 : (long)c0 * K;
  if (bias) bias += c0;
  // B fragments: lane (column n = 16 t + lr, k quad lq + 4 j) holds W[n][4 (lq + 4 j) .. + 3]
  float4 bw[NT][KQ];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
      const int n = 16 * t + lr, k = 4 * (lq + 4 * j);
      if (!WT) bw[t][j] = *reinterpret_cast<const float4*>(w + (long)n * K + k);
      else bw[t][j] = make_float4(w[(long)k * NALL + n], w[(long)(k + 1) * NALL + n], w[(long)(k + 2) * NALL + n], w[(long)(k + 3) * NALL + n]);
    }
  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = bias ? bias[16 * t + lr] : 0.f;
  const long nblocks = (rows + 15) / 16;
  for (long b = gw / NH; b < nblocks; b += (long)gridDim.x * 4 / NH) {
    const long row = b * 16 + lr < rows ? b * 16 + lr : rows - 1;   // rows past the end repeat the last one (never stored)
    float4 a[KQ];
#pragma unroll
    for (int j = 0; j < KQ; ++j) a[j] = *reinterpret_cast<const float4*>(x + row * K + 4 * (lq + 4 * j));
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KQ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av = f4c(a[j], i);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = icl_mfma_16x16x4(av, f4c(bw[t][j], i), acc[t]);
      }
    // D[row 4 lq + r][column 16 t + lr] -> LDS tile -> whole rows
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[t][r] + bv[t];
        if (act == 1) v = gelu_erf(v);
        tl[(4 * lq + r) * LDP + 16 * t + lr] = v;
      }
    ICL_WAVE_SYNC();
#pragma unroll
    for (int it = lane; it < 16 * (N / 4); it += 64) {
      const int r = it / (N / 4), q = it % (N / 4);
      if (b * 16 + r < rows)
        *reinterpret_cast<float4*>(y + (b * 16 + r) * NALL + c0 + 4 * q) = *reinterpret_cast<const float4*>(tl + r * LDP + 4 * q);
    }
    ICL_WAVE_SYNC();
  }
}

}  // namespace icl
