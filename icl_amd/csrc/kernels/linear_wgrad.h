// linear_wgrad.h — weight/bias gradient of nn.Linear for TALL activations:  dW[O][I] = sum_r G[r][O] * X[r][I],  db[O] = sum_r G[r][O]
// with r running over up to 221,184 tokens (the 48^3 x 2 token grid of SwinUNETR's first stage,
// /root/reference/code/networks/swinunetr_icl.py:703,705,812 — qkv / proj / MLPBlock linears) while O x I is only 48..768 wide.
// A library GEMM sees a 48x144 output and launches ten workgroups for a 221K-deep reduction (0.65 ms); the op is a pure
// HBM stream of G and X (170 MB -> ~35 us), so the rows are split over hundreds of waves instead:
//   wave = one 48x48 block of dW (3x3 MFMA tiles) over one slice of the rows; both MFMA operands are loaded straight from
//   HBM in their natural layout (A[o][k=row] = G[row][o], B[k=row][i] = X[row][i]: 16 consecutive floats per row group,
//   64-byte segments), every wave writes its partial block to its own slab, and a second kernel sums the slabs in a fixed
//   order (bitwise reproducible, no atomics).
#pragma once

namespace icl {

constexpr int kLwT = 3;             // MFMA tiles per block side: 48 x 48 outputs per wave
constexpr int kLwB = 16 * kLwT;
constexpr int kLwUnroll = 4;        // row groups (of 4 rows) in flight per iteration

// grid (nsplit/4, ceil(O/48), ceil(I/48)), block 256 (4 waves = 4 row slices).  slab layout: [nsplit][O*I + O].
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                           float* __restrict__ slabs, long rows, int O, int I, int nsplit) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int split = blockIdx.x * 4 + wid;
  const int o0 = blockIdx.y * kLwB, i0 = blockIdx.z * kLwB;
  const long steps = (rows + 3) / 4;                       // row groups of 4 (one MFMA k-step each)
  const long per = (steps + nsplit - 1) / nsplit;
  const long s0 = (long)split * per, s1 = (s0 + per < steps) ? s0 + per : steps;
  f32x4 acc[kLwT][kLwT];
  float bsum[kLwT];
#pragma unroll
  for (int a = 0; a < kLwT; ++a) {
    bsum[a] = 0.f;
#pragma unroll
    for (int b = 0; b < kLwT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bool ov[kLwT], iv[kLwT];
#pragma unroll
  for (int a = 0; a < kLwT; ++a) {
    ov[a] = o0 + a * 16 + lr < O;
    iv[a] = i0 + a * 16 + lr < I;
  }
  for (long s = s0; s < s1; s += kLwUnroll) {
    float av[kLwUnroll][kLwT], bv[kLwUnroll][kLwT];
#pragma unroll
    for (int u = 0; u < kLwUnroll; ++u) {
      const long r = (s + u) * 4 + lg;
      const bool rv = (s + u) < s1 && r < rows;
#pragma unroll
      for (int a = 0; a < kLwT; ++a) {
        av[u][a] = (rv && ov[a]) ? g[r * O + o0 + a * 16 + lr] : 0.f;
        bv[u][a] = (rv && iv[a]) ? x[r * I + i0 + a * 16 + lr] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < kLwUnroll; ++u)
#pragma unroll
      for (int a = 0; a < kLwT; ++a) {
        if (o0 + a * 16 >= O) continue;                    // (wave-uniform: tiles wholly outside [O] x [I] issue nothing, round 6)
        bsum[a] += av[u][a];
#pragma unroll
        for (int b = 0; b < kLwT; ++b) {
          if (i0 + b * 16 >= I) continue;
          acc[a][b] = icl_mfma_16x16x4(av[u][a], bv[u][b], acc[a][b]);
        }
      }
  }
  float* slab = slabs + (long)split * ((long)O * I + O);
#pragma unroll
  for (int a = 0; a < kLwT; ++a)
#pragma unroll
    for (int b = 0; b < kLwT; ++b) {
      const int i = i0 + b * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + a * 16 + lg * 4 + r;
        if (o < O && i < I) slab[(long)o * I + i] = acc[a][b][r];
      }
    }
  if (blockIdx.z == 0) {
#pragma unroll
    for (int a = 0; a < kLwT; ++a) {
      float v = bsum[a];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lg == 0 && ov[a]) slab[(long)O * I + o0 + a * 16 + lr] = v;
    }
  }
}

// The same reduction for CHANNEL-MAJOR operands: the weight gradient of a 1x1x1 convolution on big volumes,
//   dW[o][i] = sum_{b, v} gy[b][o][v] * x[b][i][v],  db[o] = sum gy   (x [N, I, S], gy [N, O, S], v contiguous)
// (MONAI UnetResBlock.conv3 of SwinUNETR on 48^3 / 96^3 volumes, the 1x1 heads).  The MFMA k index is only a label, so lane
// (lr, lg) takes the four voxels v0 + 4*lg + t (t = k-step): one 16-byte load per operand row and 4 k-steps, 64 contiguous
// bytes per channel row and wave — no LDS, no transposes.  S % 4 == 0.  Same slab layout / reduce kernel as above.
// grid (nsplit/4, ceil(O/48), ceil(I/48)), block 256; the (batch, voxel-group) range is split over nsplit waves.
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            float* __restrict__ slabs, int N, long S, int O, int I, int nsplit,
                                                            long g_bstride, long x_bstride, DropSpec dr) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const unsigned dseed = dr.mode ? drop_seed(dr) : 0u;      // mode 1: x is a dropped tensor (contiguous [N, I, S]) whose mask is applied on load
  const int split = blockIdx.x * 4 + wid;
  const int o0 = blockIdx.y * kLwB, i0 = blockIdx.z * kLwB;
  const long gps = S / 16;                                 // groups of 16 voxels per sample
  const long groups = gps * N + ((S % 16) ? N : 0);        // a ragged tail group per sample when S % 16 != 0
  const long gpsr = (S + 15) / 16;
  const long per = (groups + nsplit - 1) / nsplit;
  const long q0 = (long)split * per, q1 = (q0 + per < groups) ? q0 + per : groups;
  (void)gps;
  f32x4 acc[kLwT][kLwT];
  float bsum[kLwT];
#pragma unroll
  for (int a = 0; a < kLwT; ++a) {
    bsum[a] = 0.f;
#pragma unroll
    for (int b = 0; b < kLwT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bool ov[kLwT], iv[kLwT];
#pragma unroll
  for (int a = 0; a < kLwT; ++a) {
    ov[a] = o0 + a * 16 + lr < O;
    iv[a] = i0 + a * 16 + lr < I;
  }
  // Round 6: tiles of the 48 x 48 block that lie wholly outside [O] x [I] are skipped (wave-uniform tests) — the `final` convolution
  // (O = 2, I = 16) issued nine tiles' MFMAs for one, 135 us for a 127 MB stream — and two voxel groups are loaded per iteration (their
  // products stay in group order: the sums are the old ones)
  bool ot[kLwT], it[kLwT];
#pragma unroll
  for (int a = 0; a < kLwT; ++a) {
    ot[a] = o0 + a * 16 < O;
    it[a] = i0 + a * 16 < I;
  }
  for (long q = q0; q < q1; q += 2) {
    float4 av[2][kLwT], bv[2][kLwT];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long qq = q + u;
      const long b = qq / gpsr, v = (qq - b * gpsr) * 16 + lg * 4;
      const bool rv = qq < q1 && v < S;                     // S % 4 == 0: a float4 is either fully inside or fully outside
#pragma unroll
      for (int a = 0; a < kLwT; ++a) {
        av[u][a] = make_float4(0.f, 0.f, 0.f, 0.f);
        bv[u][a] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rv && ov[a]) av[u][a] = *reinterpret_cast<const float4*>(g + b * g_bstride + (long)(o0 + a * 16 + lr) * S + v);
        if (rv && iv[a]) {
          bv[u][a] = *reinterpret_cast<const float4*>(x + b * x_bstride + (long)(i0 + a * 16 + lr) * S + v);
          if (dr.ss) bv[u][a] = norm_relu4(bv[u][a], dr.ss[2 * (b * I + (i0 + a * 16 + lr))], dr.ss[2 * (b * I + (i0 + a * 16 + lr)) + 1]);
          if (dr.mode == 1) bv[u][a] = drop_apply4(bv[u][a], dseed, dr.thresh, dr.scale, (b * I + (i0 + a * 16 + lr)) * S + v);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int a = 0; a < kLwT; ++a) {
        if (!ot[a]) continue;
        bsum[a] += (av[u][a].x + av[u][a].y) + (av[u][a].z + av[u][a].w);
#pragma unroll
        for (int c = 0; c < kLwT; ++c) {
          if (!it[c]) continue;
          acc[a][c] = icl_mfma_16x16x4(av[u][a].x, bv[u][c].x, acc[a][c]);
          acc[a][c] = icl_mfma_16x16x4(av[u][a].y, bv[u][c].y, acc[a][c]);
          acc[a][c] = icl_mfma_16x16x4(av[u][a].z, bv[u][c].z, acc[a][c]);
          acc[a][c] = icl_mfma_16x16x4(av[u][a].w, bv[u][c].w, acc[a][c]);
        }
      }
  }
  float* slab = slabs + (long)split * ((long)O * I + O);
#pragma unroll
  for (int a = 0; a < kLwT; ++a)
#pragma unroll
    for (int c = 0; c < kLwT; ++c) {
      const int i = i0 + c * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + a * 16 + lg * 4 + r;
        if (o < O && i < I) slab[(long)o * I + i] = acc[a][c][r];
      }
    }
  if (blockIdx.z == 0) {
#pragma unroll
    for (int a = 0; a < kLwT; ++a) {
      float v = bsum[a];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lg == 0 && ov[a]) slab[(long)O * I + o0 + a * 16 + lr] = v;
    }
  }
}

// out[e] = sum_s slabs[s][e] for e < E (E = O*I + O; dW then db).  Block 256 = 16 consecutive elements x 16 slab lanes:
// every slab row is read as a 64-byte segment, each thread adds nsplit/16 values, the 16 partial sums are combined through
// LDS in lane order (fixed summation order -> reproducible).  grid ceil(E/16).
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                  float* __restrict__ db, long E, long EW, int nsplit) {
  __shared__ float red[16][17];
  const int le = threadIdx.x & 15, ls = threadIdx.x >> 4;
  const long e = (long)blockIdx.x * 16 + le;
  float v = 0.f;
  if (e < E) {
#pragma unroll 4
    for (int s = ls; s < nsplit; s += 16) v += slabs[(long)s * E + e];
  }
  red[ls][le] = v;
  __syncthreads();
  if (ls != 0 || e >= E) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) v += red[k][le];
  if (e < EW) dw[e] = v;
  else if (db) db[e - EW] = v;
}

}  // namespace icl
