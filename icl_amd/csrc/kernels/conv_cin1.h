// conv_cin1.h — the first convolution of the backbones (one input channel -> up to 16 output channels, 3x3x3, pad 1) on the fp32 MFMA.
//
// Reference op: nn.Conv3d(in_channels, filters[0], 3, 1, 1) inside UnetConv3 (/root/reference/code/networks/utils.py:104) as used by
// unet_3D(in_channels=1) (networks/unet_3D.py:27) on 96^3 volumes.  K = 27: the implicit-GEMM kernels pad Cin to 16 and spend 67 us
// (forward) / 196 us (weight gradient through 27 materialised planes, 190 MB) on 1.5 GFLOP; both are HBM streams of the 16-channel
// tensor (113 MB at batch 2): ~25 us.
//
// One workgroup (4 waves) owns a (z, 8 y rows, all x) slab; the x halo of the slab (3 x 10 x (W + 2) floats, zero-padded) sits in LDS.
// v_mfma_f32_16x16x4_f32 (lane l = (lr = l & 15, lg = l >> 4): A[row lr][k lg], B[k lg][col lr], D[row 4 lg + r][col lr] in register r);
// the k index of a step is only a label, so a lane's float4 of FOUR CONSECUTIVE x positions feeds four steps (step j: position 4 lg + j of
// the group of 16) and no register is transposed:
//   weight gradient  dW[cout][tap] = sum_p dY[cout][p] x[p + tap]: A = dY (rows = couts, one float4 global load per lane and 16
//     positions), B = x[p + tap] (cols = taps 0..15 / 16..31, one ds_read_b32 per lane and step): 8 MFMAs per 16 positions; the
//     accumulators (2 x 4 registers) live for the whole launch, are summed over the workgroup's waves through LDS and written as one
//     [16][32] partial per workgroup; conv_cin1_wgrad_reduce_kernel adds the partials in a fixed order.
//   forward  y[cout][p] = b[cout] + sum_tap w[cout][tap] x[p + tap]: A = w (rows = couts, k = taps: 7 steps, the 28th tap is zero),
//     B = x[p + tap] (cols = 16 positions), D[cout 4 lg + r][position lr]: 7 MFMAs per 16 positions, 4 stores of 64 bytes per lane group.
#pragma once

namespace icl {

constexpr int kC1Rows = 8, kC1Threads = 256;

struct Cin1Geom {
  int N, Cout, D, H, W, nty;        // nty = ceil(H / 8); tiles = N * D * nty
  long x_bstride, y_bstride;        // elements between samples of x ([N][1][D][H][W]) and of y / dY ([N][Cout][D][H][W])
  int ntiles;
};

// zero-padded halo of tile (n, z, y0): rows z-1..z+1, y0-1..y0+8, x -1..W
__device__ __forceinline__ void c1_stage_halo(float* xs, const float* __restrict__ x, const Cin1Geom& g, int n, int z, int y0) {
  const int PW = g.W + 2, PR = kC1Rows + 2;
  const long HW = (long)g.H * g.W;
  const float* xb = x + (long)n * g.x_bstride;
  for (int it = threadIdx.x; it < 3 * PR * PW; it += kC1Threads) {
    const int px = it % PW, row = it / PW, py = row % PR, pz = row / PR;
    const int gz = z - 1 + pz, gy = y0 - 1 + py, gx = px - 1;
    float v = 0.f;
    if ((unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W) v = xb[gz * HW + (long)gy * g.W + gx];
    xs[it] = v;
  }
}

// grid = workgroups (each walks tiles wg, wg + gridDim.x, ...); LDS = 3 * 10 * (W + 2) floats (>= 4 * 2 * 256 floats for the final sum).
// part: [gridDim.x][16][32] floats.  W % 4 == 0.
__global__ __launch_bounds__(kC1Threads) void conv_cin1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                     float* __restrict__ part, Cin1Geom g) {
  ICL_DYN_LDS(float, xs);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int PW = g.W + 2, PR = kC1Rows + 2;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  // tap of this lane's B column in the two tap blocks, as an offset into the halo (taps >= 27: a zero is multiplied instead)
  int toff[2];
  bool tok[2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {
    const int tap = 16 * tb + lr;
    tok[tb] = tap < 27;
    const int t = tok[tb] ? tap : 0;
    toff[tb] = (t / 9) * PR * PW + ((t / 3) % 3) * PW + t % 3;
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    const int n = tile / (g.D * g.nty), rem = tile % (g.D * g.nty), z = rem / g.nty, y0 = (rem % g.nty) * kC1Rows;
    __syncthreads();                                      // the previous tile's halo has been read by everyone
    c1_stage_halo(xs, x, g, n, z, y0);
    __syncthreads();
    const float* gb = gy + (long)n * g.y_bstride + (long)lr * DHW + (long)z * HW;
    // wave wid owns rows 2 wid, 2 wid + 1 of the tile; groups of 16 x positions
    for (int rr = 0; rr < 2; ++rr) {
      const int ty = 2 * wid + rr, yy = y0 + ty;
      if (yy >= g.H) break;                               // wave-uniform
      const float* grow = gb + (long)yy * g.W;
      const float* xrow = xs + (ty * PW);                 // halo row of tap (0, 0, 0) for output row ty
      for (int x0 = 0; x0 < g.W; x0 += 16) {
        const int xl = x0 + 4 * lg;                       // the lane's four positions xl .. xl + 3
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lr < g.Cout && xl < g.W) a = *reinterpret_cast<const float4*>(grow + xl);
        const float av[4] = {a.x, a.y, a.z, a.w};
        const bool in = xl < g.W;                         // positions beyond W (a partial last group) contribute nothing: a = 0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int tb = 0; tb < 2; ++tb) {
            const float b = (tok[tb] && in) ? xrow[toff[tb] + xl + j] : 0.f;
            acc[tb] = icl_mfma_16x16x4(av[j], b, acc[tb]);
          }
        }
      }
    }
  }
  // sum over the four waves: lane-private slots, wave 0 adds
  __syncthreads();
  float* red = xs;
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wid * 2 + tb) * 4 + r) * 64 + lane] = acc[tb][r];
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * 2 + tb) * 4 + r) * 64 + lane];
        // D[row = cout 4 lg + r][col = tap 16 tb + lr]
        part[(long)blockIdx.x * 512 + (4 * lg + r) * 32 + 16 * tb + lr] = s;
      }
  }
}

// gw[cout][tap] (the nn.Conv3d weight gradient [Cout][1][3][3][3]) = sum over the partials, fixed order: 16 workgroups x 32 outputs,
// eight slices of the partial list per output (thread = (output, slice)), the slices added through LDS.  grid 16, block 256.
__global__ __launch_bounds__(256) void conv_cin1_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int nparts, int cout) {
  __shared__ float sl[8][32];
  const int o = threadIdx.x & 31, slice = threadIdx.x >> 5, e = blockIdx.x * 32 + o;
  float s = 0.f;
#pragma unroll 8
  for (int p = slice; p < nparts; p += 8) s += part[(long)p * 512 + e];
  sl[slice][o] = s;
  __syncthreads();
  if (slice == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sl[k][o];
    const int co = e >> 5, tap = e & 31;
    if (co < cout && tap < 27) gw[co * 27 + tap] = t;
  }
}

// forward.  w: the nn.Conv3d weight [Cout][1][27]; grid as above; LDS = 3 * 10 * (W + 2) floats.  W % 4 == 0 is not required here.
__global__ __launch_bounds__(kC1Threads) void conv_cin1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, float* __restrict__ y, Cin1Geom g) {
  ICL_DYN_LDS(float, xs);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int PW = g.W + 2, PR = kC1Rows + 2;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  // A[row = cout lr][k = tap 4 s + lg] of step s, and that tap's halo offset (B[k][col = position lr])
  float wa[7];
  int toff[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int tap = 4 * s + lg;
    wa[s] = (tap < 27 && lr < g.Cout) ? w[lr * 27 + tap] : 0.f;
    const int t = tap < 27 ? tap : 0;
    toff[s] = (t / 9) * PR * PW + ((t / 3) % 3) * PW + t % 3;
  }
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = (bias && 4 * lg + r < g.Cout) ? bias[4 * lg + r] : 0.f;
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    const int n = tile / (g.D * g.nty), rem = tile % (g.D * g.nty), z = rem / g.nty, y0 = (rem % g.nty) * kC1Rows;
    __syncthreads();
    c1_stage_halo(xs, x, g, n, z, y0);
    __syncthreads();
    float* yb = y + (long)n * g.y_bstride + (long)z * HW;
    for (int rr = 0; rr < 2; ++rr) {
      const int ty = 2 * wid + rr, yy = y0 + ty;
      if (yy >= g.H) break;
      const float* xrow = xs + ty * PW;
      for (int x0 = 0; x0 < g.W; x0 += 16) {
        const int xp = x0 + lr;                           // the lane's B column
        const bool in = xp < g.W;
        f32x4 acc = f32x4{bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
        for (int s = 0; s < 7; ++s) acc = icl_mfma_16x16x4(wa[s], in ? xrow[toff[s] + xp] : 0.f, acc);
        // D[cout 4 lg + r][position lr]
        if (in) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * lg + r < g.Cout) yb[(long)(4 * lg + r) * DHW + (long)yy * g.W + xp] = acc[r];
        }
      }
    }
  }
}

}  // namespace icl
