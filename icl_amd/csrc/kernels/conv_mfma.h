// conv_mfma.h — dense 3-D convolution (kernel 3^3 pad 1, or 1^3) as an LDS-tiled implicit GEMM on the
// f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, bitwise a k-ordered fmaf chain).
//
// Reference op: nn.Conv3d(k=3, stride 1, pad 1, bias) inside UnetConv3
// (/root/reference/code/networks/utils.py:104,107) and the 1x1x1 convs
// (networks/unet_3D_icl.py:65 final, :178 proj_layers, :196 attn_convs1, :327 pointwise).
//
// Data layout: activations NCDHW fp32 (PyTorch contiguous; a batch stride lets a launch read or
// write a channel slice of a concat buffer).  Weights are re-packed once per step by
// pack_weights_kernel into Wp[tap][CinP][CoutP] (CinP % 4 == 0, CoutP % 16 == 0, zero padded):
//   forward : Wp[tap][ci][co]       = W[co][ci][tap]
//   dgrad   : Wp[T-1-tap][co][ci]   = W[co][ci][tap]   (dX = conv(dY, flipped/transposed W): same kernel)
// so weight staging is a coalesced row copy and the MFMA B operand is bank-conflict free.
//
// forward / dgrad kernel (one workgroup = one TZxTYxTX output tile x NB output channels):
//   GEMM view  M = voxels (16 consecutive tile voxels per MFMA), N = Cout (16 per MFMA),
//              K = taps x Cin, walked tap-major in steps of 4 input channels.
//   A[v][k]  = Xs[cin plane k][voxel v + tap offset]   (input halo tile staged in LDS, zero filled)
//   B[k][n]  = Ws[tap*KC + k][n]
//   LDS plane stride PS == 16 (mod 32): lanes 0-15 (voxels) and 16-31 (next cin plane) hit disjoint banks.
// wgrad kernel (one workgroup = 16 output channels x 16 input channels x all taps, loops over tiles):
//   GEMM view  M = Cout (16), N = Cin (16) per tap, K = voxels in steps of 4 consecutive x.
//   A[co][v] = Gs[co][v] (dY tile, zero where outside the volume), B[v][ci] = Xs[ci][v + tap offset]
//   taps are dealt round-robin to the waves; partial dW is atomically added into packed gWp.
//
// Algorithmic HBM bytes: fwd 4*(I+O+W), dgrad 4*(O+I+W), wgrad 4*(O+I+W).
#pragma once

namespace icl {

struct ConvGeom {
  int Cin, Cout;    // logical channels read / written by this launch
  int CinP, CoutP;  // packed weight extents
  int D, H, W;
  int TZ, TY, TX;  // output tile
  int ntz, nty, ntx;
  int KC;  // input channels per LDS chunk (multiple of 4)
  int PS;  // LDS plane stride in floats
  long x_bstride, y_bstride;  // batch strides in elements (channel stride is D*H*W)
};

// Wp[tap'][k][n] (zero padded) from W[co][ci][tap]; mode 0 = forward (k=ci,n=co), 1 = dgrad (k=co,n=ci, tap flipped)
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                           int T, int KP, int NP, int mode) {
  const long total = (long)T * KP * NP;
  const int K = mode ? Cout : Cin, N = mode ? Cin : Cout;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e % NP);
    const long t = e / NP;
    const int k = (int)(t % KP);
    const int tap = (int)(t / KP);
    float v = 0.f;
    if (k < K && n < N) {
      v = mode ? w[((long)k * Cin + n) * T + (T - 1 - tap)] : w[((long)n * Cin + k) * T + tap];
    }
    wp[e] = v;
  }
}

// gW[co][ci][tap] = gWp[tap][ci][co]
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ gwp, float* __restrict__ gw, int Cout, int Cin,
                                                           int T, int CinP, int CoutP) {
  const long total = (long)Cout * Cin * T;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(e % T);
    const long t = e / T;
    const int ci = (int)(t % Cin);
    const int co = (int)(t / Cin);
    gw[e] = gwp[((long)tap * CinP + ci) * CoutP + co];
  }
}

// Stage the zero-filled input halo tile of channels [c0, c0+KC) into Xs[c][PS]; rows are (c,pz,py), 32 lanes per row.
template <int KS, int NT>
__device__ __forceinline__ void stage_halo(float* Xs, const float* __restrict__ xb, int c0, const ConvGeom& g, int z0, int y0, int x0) {
  constexpr int PAD = KS / 2;
  const int PZ = g.TZ + KS - 1, PY = g.TY + KS - 1, PX = g.TX + KS - 1;
  const int rows = g.KC * PZ * PY;
  const int px = threadIdx.x & 31;
  const int gx = x0 - PAD + px;
  const bool xok = (px < PX) && gx >= 0 && gx < g.W;
  const long HW = (long)g.H * g.W;
  for (int rr = threadIdx.x >> 5; rr < rows; rr += NT / 32) {
    const int c = rr / (PZ * PY);
    const int rem = rr - c * (PZ * PY);
    const int pz = rem / PY, py = rem - pz * PY;
    const int gz = z0 - PAD + pz, gy = y0 - PAD + py;
    float v = 0.f;
    if (xok && (c0 + c) < g.Cin && gz >= 0 && gz < g.D && gy >= 0 && gy < g.H)
      v = xb[(long)(c0 + c) * g.D * HW + gz * HW + (long)gy * g.W + gx];
    if (px < PX) Xs[c * g.PS + (pz * PY + py) * PX + px] = v;
  }
}

template <int KS, int NBT, int MV, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void conv3d_mfma_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                                     ConvGeom g) {
  constexpr int T = KS * KS * KS;
  constexpr int NB = NBT * 16;
  constexpr int NBP = NB + (NB > 16 ? 16 : 0);
  constexpr int NT = WAVES * 64;
  ICL_DYN_LDS(float, lds);
  float* Xs = lds;
  float* Ws = lds + g.KC * g.PS;
  const int PY = g.TY + KS - 1, PX = g.TX + KS - 1;
  const int bt = blockIdx.x;
  const int x0 = (bt % g.ntx) * g.TX;
  const int y0 = ((bt / g.ntx) % g.nty) * g.TY;
  const int z0 = (bt / (g.ntx * g.nty)) * g.TZ;
  const int n0 = blockIdx.y * NB;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const float* xb = x + (long)blockIdx.z * g.x_bstride;
  float* yb = y + (long)blockIdx.z * g.y_bstride;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  const int MT = g.TZ * g.TY * g.TX;

  int vbase[MV];
#pragma unroll
  for (int m = 0; m < MV; ++m) {
    int vt = (wid * MV + m) * 16 + lr;
    if (vt >= MT) vt = MT - 1;
    const int tx = vt % g.TX, t2 = vt / g.TX;
    const int ty = t2 % g.TY, tz = t2 / g.TY;
    vbase[m] = (tz * PY + ty) * PX + tx + lq * g.PS;
  }
  f32x4 acc[MV][NBT];
#pragma unroll
  for (int m = 0; m < MV; ++m)
#pragma unroll
    for (int j = 0; j < NBT; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < g.CinP; c0 += g.KC) {
    __syncthreads();
    stage_halo<KS, NT>(Xs, xb, c0, g, z0, y0, x0);
    {
      const int total = T * g.KC * NB;
      for (int e = threadIdx.x; e < total; e += NT) {
        const int n = e % NB, row = e / NB;
        const int tap = row / g.KC, c = row - tap * g.KC;
        float v = 0.f;
        if (n0 + n < g.CoutP && c0 + c < g.CinP) v = wp[((long)tap * g.CinP + c0 + c) * g.CoutP + n0 + n];
        Ws[row * NBP + n] = v;
      }
    }
    __syncthreads();
    for (int tap = 0; tap < T; ++tap) {
      const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
      const int tapoff = (dz * PY + dy) * PX + dx;
      for (int cc = 0; cc < g.KC; cc += 4) {
        const int krow = tap * g.KC + cc + lq;
        float b[NBT];
#pragma unroll
        for (int j = 0; j < NBT; ++j) b[j] = Ws[krow * NBP + j * 16 + lr];
#pragma unroll
        for (int m = 0; m < MV; ++m) {
          const float a = Xs[vbase[m] + tapoff + cc * g.PS];
#pragma unroll
          for (int j = 0; j < NBT; ++j) acc[m][j] = icl_mfma_16x16x4(a, b[j], acc[m][j]);
        }
      }
    }
  }

  // epilogue: lane holds, per (m, j), rows vt = group*16 + lq*4 + r (r = 0..3) of column co = n0 + j*16 + lr
  const bool vec = ((g.TX & 3) == 0) && ((g.W & 3) == 0);
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int co = n0 + j * 16 + lr;
    if (co >= g.Cout) continue;
    const float bv = bias ? bias[co] : 0.f;
    float* yc = yb + (long)co * DHW;
#pragma unroll
    for (int m = 0; m < MV; ++m) {
      const int vt0 = (wid * MV + m) * 16 + lq * 4;
      if (vt0 >= MT) continue;
      if (vec) {
        const int tx = vt0 % g.TX, t2 = vt0 / g.TX;
        const int ty = t2 % g.TY, tz = t2 / g.TY;
        const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
        if (gz < g.D && gy < g.H && gx < g.W) {
          float4 o = make_float4(acc[m][j][0] + bv, acc[m][j][1] + bv, acc[m][j][2] + bv, acc[m][j][3] + bv);
          *reinterpret_cast<float4*>(yc + gz * HW + (long)gy * g.W + gx) = o;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int vt = vt0 + r;
          if (vt >= MT) continue;
          const int tx = vt % g.TX, t2 = vt / g.TX;
          const int ty = t2 % g.TY, tz = t2 / g.TY;
          const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
          if (gz < g.D && gy < g.H && gx < g.W) yc[gz * HW + (long)gy * g.W + gx] = acc[m][j][r] + bv;
        }
      }
    }
  }
}

// dW partials: grid.x = spatial split, grid.y = (CoutP/16) * ceil(CinP/16), grid.z = batch.
// Requires TX % 4 == 0 and KC == 16.  Gs pitch MTP and plane stride PS are == 2 (mod 32).
template <int KS, int NTW, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void conv3d_mfma_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                       float* __restrict__ gwp, ConvGeom g, int MTP) {
  constexpr int T = KS * KS * KS;
  constexpr int NT = WAVES * 64;
  ICL_DYN_LDS(float, lds);
  float* Xs = lds;                // [16][PS]
  float* Gs = lds + 16 * g.PS;    // [16][MTP]
  const int PY = g.TY + KS - 1, PX = g.TX + KS - 1;
  const int ncin = (g.CinP + 15) / 16;
  const int co0 = (blockIdx.y / ncin) * 16;
  const int c0 = (blockIdx.y % ncin) * 16;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const float* xb = x + (long)blockIdx.z * g.x_bstride;
  const float* gb = gy + (long)blockIdx.z * g.y_bstride;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  const int MT = g.TZ * g.TY * g.TX;
  const int ntiles = g.ntz * g.nty * g.ntx;

  int tapoff[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int tap = wid + t * WAVES;
    const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
    tapoff[t] = (dz * PY + dy) * PX + dx;
  }
  f32x4 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int bt = blockIdx.x; bt < ntiles; bt += gridDim.x) {
    const int x0 = (bt % g.ntx) * g.TX;
    const int y0 = ((bt / g.ntx) % g.nty) * g.TY;
    const int z0 = (bt / (g.ntx * g.nty)) * g.TZ;
    __syncthreads();
    stage_halo<KS, NT>(Xs, xb, c0, g, z0, y0, x0);
    for (int e = threadIdx.x; e < 16 * MT; e += NT) {
      const int vt = e % MT, co = e / MT;
      const int tx = vt % g.TX, t2 = vt / g.TX;
      const int ty = t2 % g.TY, tz = t2 / g.TY;
      const int gz = z0 + tz, gyy = y0 + ty, gx = x0 + tx;
      float v = 0.f;
      if (co0 + co < g.Cout && gz < g.D && gyy < g.H && gx < g.W) v = gb[(long)(co0 + co) * DHW + gz * HW + (long)gyy * g.W + gx];
      Gs[co * MTP + vt] = v;
    }
    __syncthreads();
    for (int k0 = 0; k0 < MT; k0 += 4) {
      const int tx = k0 % g.TX, t2 = k0 / g.TX;
      const int ty = t2 % g.TY, tz = t2 / g.TY;
      const int vb = (tz * PY + ty) * PX + tx + lq + lr * g.PS;
      const float a = Gs[lr * MTP + k0 + lq];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        if (wid + t * WAVES < T) {
          const float b = Xs[vb + tapoff[t]];
          acc[t] = icl_mfma_16x16x4(a, b, acc[t]);
        }
      }
    }
  }
  // D[row = co0 + lq*4 + r][col = c0 + lr]
  const int ci = c0 + lr;
  if (ci < g.CinP) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int tap = wid + t * WAVES;
      if (tap >= T) continue;
      float* dst = gwp + ((long)tap * g.CinP + ci) * g.CoutP + co0 + lq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(dst + r, acc[t][r]);
    }
  }
}

}  // namespace icl
