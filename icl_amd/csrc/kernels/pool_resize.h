// pool_resize.h — MaxPool3d(2) / MaxPool2d(2), (tri/bi)linear resize and strided row copy.
//
// Reference ops: nn.MaxPool3d(kernel_size=2) (/root/reference/code/networks/unet_3D_icl.py:41-53), nn.MaxPool2d(2)
// (networks/unet_icl.py:64), nn.Upsample(scale_factor=2, mode='trilinear') in UnetUp3_CT (networks/utils.py:264),
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) in the 2D UpBlock (networks/unet_icl.py:84-85)
// and F.interpolate(size=..., mode='trilinear'|'bilinear') in the losses (utils/losses.py:245,263,281,292).
// 2-D tensors are handled as D = 1 volumes.  All HBM-bound; x is the fastest index so loads/stores coalesce.
#pragma once

namespace icl {

// ---- max pooling, window pd x 2 x 2 (pd = 2: MaxPool3d(2); pd = 1: MaxPool2d(2) on a D=1 volume), stride = window.
// idx = argmax in (dz,dy,dx) scan order with strict '>' so the first maximum wins, as ATen's max_pool does.
// ss != nullptr: x is a RAW convolution output whose InstanceNorm + ReLU is applied on load (norm.h norm_finalize_stats_kernel): the
// maximum and its index are those of the normalised tensor (ReLU ties resolve to the first maximum, as on a materialised tensor).
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           unsigned char* __restrict__ idx, long NC, int Do, int Ho, int Wo, int pd,
                                                           const float* __restrict__ ss) {
  const long total = NC * Do * Ho * Wo;
  const int H = Ho * 2, W = Wo * 2;
  const int nk = 4 * pd;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(o % Wo);
    long t = o / Wo;
    const int oy = (int)(t % Ho);
    t /= Ho;
    const int oz = (int)(t % Do);
    const long nc = t / Do;
    const float* p = x + ((nc * (Do * pd) + oz * pd) * H + oy * 2) * (long)W + ox * 2;
    const float sc = ss ? ss[2 * nc] : 1.f, sh = ss ? ss[2 * nc + 1] : 0.f;
    float best = ss ? norm_relu1(p[0], sc, sh) : p[0];
    int bi = 0;
    for (int k = 1; k < nk; ++k) {
      float v = p[((k >> 2) * H + ((k >> 1) & 1)) * (long)W + (k & 1)];
      if (ss) v = norm_relu1(v, sc, sh);
      if (v > best) { best = v; bi = k; }
    }
    y[o] = best;
    idx[o] = (unsigned char)bi;
  }
}

// The same for TWO x-neighbouring outputs per thread (round 6; Wo even, rows 16-byte aligned): one 16-byte load per window row instead
// of four 4-byte ones (conv1's 96^3 output, batch 2: a 113 MB read at the head of every forward pass, 40 us with the scalar form).
// Same scan order per window, same maxima and indices.
__global__ __launch_bounds__(256) void maxpool2_fwd_x2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              unsigned char* __restrict__ idx, long NC, int Do, int Ho, int Wo, int pd,
                                                              const float* __restrict__ ss) {
  const int Wo2 = Wo >> 1;
  const long total = NC * Do * Ho * Wo2;
  const int H = Ho * 2, W = Wo * 2;
  const int nk = 4 * pd;
  for (long o2 = (long)blockIdx.x * blockDim.x + threadIdx.x; o2 < total; o2 += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(o2 % Wo2) * 2;
    long t = o2 / Wo2;
    const int oy = (int)(t % Ho);
    t /= Ho;
    const int oz = (int)(t % Do);
    const long nc = t / Do;
    const long o = ((nc * Do + oz) * Ho + oy) * (long)Wo + ox;
    const float* p = x + ((nc * (Do * pd) + oz * pd) * H + oy * 2) * (long)W + ox * 2;
    const float sc = ss ? ss[2 * nc] : 1.f, sh = ss ? ss[2 * nc + 1] : 0.f;
    float b0 = 0.f, b1 = 0.f;
    int i0 = 0, i1 = 0;
    for (int k = 0; k < nk; k += 2) {
      float4 v = *reinterpret_cast<const float4*>(p + ((k >> 2) * H + ((k >> 1) & 1)) * (long)W);
      if (ss) { v.x = norm_relu1(v.x, sc, sh); v.y = norm_relu1(v.y, sc, sh); v.z = norm_relu1(v.z, sc, sh); v.w = norm_relu1(v.w, sc, sh); }
      if (k == 0) { b0 = v.x; b1 = v.z; }
      else {
        if (v.x > b0) { b0 = v.x; i0 = k; }
        if (v.z > b1) { b1 = v.z; i1 = k; }
      }
      if (v.y > b0) { b0 = v.y; i0 = k + 1; }
      if (v.w > b1) { b1 = v.w; i1 = k + 1; }
    }
    *reinterpret_cast<float2*>(y + o) = make_float2(b0, b1);
    idx[o] = (unsigned char)i0;
    idx[o + 1] = (unsigned char)i1;
  }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                           float* __restrict__ gx, long NC, int Do, int Ho, int Wo, int pd) {
  const long total = NC * Do * Ho * Wo;
  const int H = Ho * 2, W = Wo * 2;
  const int nk = 4 * pd;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(o % Wo);
    long t = o / Wo;
    const int oy = (int)(t % Ho);
    t /= Ho;
    const int oz = (int)(t % Do);
    const long nc = t / Do;
    float* p = gx + ((nc * (Do * pd) + oz * pd) * H + oy * 2) * (long)W + ox * 2;
    const float g = gy[o];
    const int bi = idx[o];
    for (int k = 0; k < nk; k += 2) {
      float2 v = make_float2(bi == k ? g : 0.f, bi == k + 1 ? g : 0.f);
      *reinterpret_cast<float2*>(p + ((k >> 2) * H + ((k >> 1) & 1)) * (long)W) = v;
    }
  }
}

// gx = add + scatter(gy): the gradient of a tensor that feeds BOTH a max-pool and a skip connection (conv1 .. conv4 of the 3D U-Net,
// /root/reference/code/networks/unet_3D_icl.py:100-116: `maxpool(conv)` and `up_concat(conv, .)`) in one pass — autograd would run the
// pooling backward (a full write of gx) and then add the skip gradient (two full reads, one write).  `add` is the skip gradient:
// [N][C] planes of the pooled-from extent, batches add_bstride apart (a channel slice of the concat gradient).
__global__ __launch_bounds__(256) void maxpool2_bwd_add_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                               const float* __restrict__ add, float* __restrict__ gx, long NC, int C, int Do,
                                                               int Ho, int Wo, int pd, long add_bstride) {
  const long total = NC * Do * Ho * Wo;
  const int H = Ho * 2, W = Wo * 2;
  const long plane = (long)Do * pd * H * W;
  const int nk = 4 * pd;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(o % Wo);
    long t = o / Wo;
    const int oy = (int)(t % Ho);
    t /= Ho;
    const int oz = (int)(t % Do);
    const long nc = t / Do;
    const long in_plane = ((long)(oz * pd) * H + oy * 2) * (long)W + ox * 2;
    float* p = gx + nc * plane + in_plane;
    const float* a = add + (nc / C) * add_bstride + (nc % C) * plane + in_plane;
    const float g = gy[o];
    const int bi = idx[o];
    for (int k = 0; k < nk; k += 2) {
      const long off = ((k >> 2) * H + ((k >> 1) & 1)) * (long)W;
      const float2 s = *reinterpret_cast<const float2*>(a + off);
      *reinterpret_cast<float2*>(p + off) = make_float2(s.x + (bi == k ? g : 0.f), s.y + (bi == k + 1 ? g : 0.f));
    }
  }
}

// The same for TWO x-neighbouring pooled outputs per thread (round 6; Wo even, rows 16-byte aligned): 16-byte loads and stores of the
// skip gradient and of gx instead of 8-byte ones.  The pass is an HBM stream of 2.25 x the tensor (conv1 of the U-Net, batch 2: 254 MB)
// on the step's critical path (the first encoder block's backward has nothing beside it): 100 us with the 8-byte form.  Same values.
__global__ __launch_bounds__(256) void maxpool2_bwd_add_x2_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                                  const float* __restrict__ add, float* __restrict__ gx, long NC, int C, int Do,
                                                                  int Ho, int Wo, int pd, long add_bstride) {
  const int Wo2 = Wo >> 1;
  const long total = NC * Do * Ho * Wo2;
  const int H = Ho * 2, W = Wo * 2;
  const long plane = (long)Do * pd * H * W;
  const int nk = 4 * pd;
  for (long o2 = (long)blockIdx.x * blockDim.x + threadIdx.x; o2 < total; o2 += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(o2 % Wo2) * 2;
    long t = o2 / Wo2;
    const int oy = (int)(t % Ho);
    t /= Ho;
    const int oz = (int)(t % Do);
    const long nc = t / Do;
    const long o = ((nc * Do + oz) * Ho + oy) * (long)Wo + ox;
    const long in_plane = ((long)(oz * pd) * H + oy * 2) * (long)W + ox * 2;
    float* p = gx + nc * plane + in_plane;
    const float* a = add + (nc / C) * add_bstride + (nc % C) * plane + in_plane;
    const float2 g = *reinterpret_cast<const float2*>(gy + o);
    const int b0 = idx[o], b1 = idx[o + 1];
    for (int k = 0; k < nk; k += 2) {
      const long off = ((k >> 2) * H + ((k >> 1) & 1)) * (long)W;
      const float4 s = *reinterpret_cast<const float4*>(a + off);
      *reinterpret_cast<float4*>(p + off) = make_float4(s.x + (b0 == k ? g.x : 0.f), s.y + (b0 == k + 1 ? g.x : 0.f),
                                                        s.z + (b1 == k ? g.y : 0.f), s.w + (b1 == k + 1 ? g.y : 0.f));
    }
  }
}

// ---- linear interpolation taps along one axis, ATen's area_pixel_compute_source_index:
//   align_corners=False: src = rscale*(dst+0.5)-0.5 clamped at 0, rscale = in/out
//   align_corners=True : src = rscale*dst,                         rscale = (in-1)/(out-1)  (0 when out == 1)
//   i0 = floor(src), i1 = i0 + (i0 < in-1), l1 = src-i0, l0 = 1-l1
struct LinTap { int i0, i1; float l0, l1; };
__device__ __forceinline__ LinTap lin_tap(int dst, float rscale, int in_size, int align) {
  float src = align ? rscale * (float)dst : rscale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  LinTap t;
  t.i0 = (int)src;
  if (t.i0 > in_size - 1) t.i0 = in_size - 1;
  t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

// y[n][c] planes are Do*Ho*Wo apart; batches y_bstride apart (lets the caller write into the
// channel slice of a concat buffer).  x is dense [N][C][Di][Hi][Wi].
__global__ __launch_bounds__(256) void trilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C,
                                                            int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                            float rz, float ry, float rx, long y_bstride, int align) {
  const long per_n = (long)C * Do * Ho * Wo;
  const long total = (long)N * per_n;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(o % Wo);
    long t = o / Wo;
    const int oy = (int)(t % Ho);
    t /= Ho;
    const int oz = (int)(t % Do);
    t /= Do;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    const LinTap tz = lin_tap(oz, rz, Di, align), ty = lin_tap(oy, ry, Hi, align), tx = lin_tap(ox, rx, Wi, align);
    const float* p = x + ((long)n * C + c) * Di * Hi * Wi;
    const long z0 = (long)tz.i0 * Hi * Wi, z1 = (long)tz.i1 * Hi * Wi;
    const long y0 = (long)ty.i0 * Wi, y1 = (long)ty.i1 * Wi;
    const float v =
        tz.l0 * (ty.l0 * (tx.l0 * p[z0 + y0 + tx.i0] + tx.l1 * p[z0 + y0 + tx.i1]) +
                 ty.l1 * (tx.l0 * p[z0 + y1 + tx.i0] + tx.l1 * p[z0 + y1 + tx.i1])) +
        tz.l1 * (ty.l0 * (tx.l0 * p[z1 + y0 + tx.i0] + tx.l1 * p[z1 + y0 + tx.i1]) +
                 ty.l1 * (tx.l0 * p[z1 + y1 + tx.i0] + tx.l1 * p[z1 + y1 + tx.i1]));
    y[(long)n * y_bstride + (((long)c * Do + oz) * Ho + oy) * Wo + ox] = v;
  }
}

// Same interpolation, four adjacent ox per thread (Wo % 4 == 0, 16-byte aligned rows) and 32-bit index
// arithmetic (host guarantees the quad count fits): the scalar kernel above spends most of its time in the
// five 64-bit div/mod pairs per output voxel, here they are five 32-bit ones per four voxels and the store
// is one float4.
__global__ __launch_bounds__(256) void trilinear_fwd_quad_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C,
                                                                 int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                                 float rz, float ry, float rx, long y_bstride, int align) {
  const unsigned wq = (unsigned)Wo >> 2;
  const unsigned total = (unsigned)N * C * Do * Ho * wq;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const unsigned xq = e % wq;
    unsigned t = e / wq;
    const int oy = (int)(t % (unsigned)Ho);
    t /= (unsigned)Ho;
    const int oz = (int)(t % (unsigned)Do);
    t /= (unsigned)Do;
    const int c = (int)(t % (unsigned)C);
    const int n = (int)(t / (unsigned)C);
    const LinTap tz = lin_tap(oz, rz, Di, align), ty = lin_tap(oy, ry, Hi, align);
    const float* p = x + ((long)n * C + c) * Di * Hi * Wi;
    const float* p00 = p + (long)tz.i0 * Hi * Wi + (long)ty.i0 * Wi;
    const float* p01 = p + (long)tz.i0 * Hi * Wi + (long)ty.i1 * Wi;
    const float* p10 = p + (long)tz.i1 * Hi * Wi + (long)ty.i0 * Wi;
    const float* p11 = p + (long)tz.i1 * Hi * Wi + (long)ty.i1 * Wi;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const LinTap tx = lin_tap((int)(xq * 4) + k, rx, Wi, align);
      v[k] = tz.l0 * (ty.l0 * (tx.l0 * p00[tx.i0] + tx.l1 * p00[tx.i1]) + ty.l1 * (tx.l0 * p01[tx.i0] + tx.l1 * p01[tx.i1])) +
             tz.l1 * (ty.l0 * (tx.l0 * p10[tx.i0] + tx.l1 * p10[tx.i1]) + ty.l1 * (tx.l0 * p11[tx.i0] + tx.l1 * p11[tx.i1]));
    }
    *reinterpret_cast<float4*>(y + (long)n * y_bstride + (((long)c * Do + oz) * Ho + oy) * Wo + xq * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---- exact 2x upsampling (align_corners = False): nn.Upsample(scale_factor=(2,2,2), mode='trilinear') of UnetUp3_CT
// (networks/utils.py:262-276).  src = (dst + 0.5)/2 - 0.5, so along each axis
//   out[2i]   = 0.25 x[i-1] + 0.75 x[i]   (i = 0: x[0] exactly — ATen clamps src to 0, l1 = 0)
//   out[2i+1] = 0.75 x[i]   + 0.25 x[min(i+1, n-1)]
// One thread produces a 2 x 2 x 4 output block (two input voxels along x) from a 3 x 3 x 4 input neighbourhood, separably
// (x, then y, then z — the nesting of the generic kernel): 36 loads and ~150 FMAs per 16 outputs instead of 128 loads and five
// div/mod chains; four 16-byte stores.  Wi % 2 == 0, 16-byte aligned output rows.
__device__ __forceinline__ float up2_even(float prev, float cur, bool first) { return first ? cur : 0.25f * prev + 0.75f * cur; }
__device__ __forceinline__ float up2_odd(float cur, float next) { return 0.75f * cur + 0.25f * next; }

__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int Di,
                                                             int Hi, int Wi, long y_bstride, const float* __restrict__ ss) {
  const unsigned wq = (unsigned)Wi >> 1;
  const unsigned total = (unsigned)N * C * Di * Hi * wq;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const unsigned xq = e % wq;
    unsigned t = e / wq;
    const int iy = (int)(t % (unsigned)Hi);
    t /= (unsigned)Hi;
    const int iz = (int)(t % (unsigned)Di);
    t /= (unsigned)Di;
    const int c = (int)(t % (unsigned)C);
    const int n = (int)(t / (unsigned)C);
    const float* p = x + ((long)n * C + c) * Di * Hi * Wi;
    const float sc = ss ? ss[2 * (n * C + c)] : 1.f, sh = ss ? ss[2 * (n * C + c) + 1] : 0.f;      // deferred InstanceNorm + ReLU of x
    const int ix = (int)xq * 2;
    const int xa = ix > 0 ? ix - 1 : 0, xd = ix + 2 < Wi ? ix + 2 : Wi - 1;
    float r[3][3][4];   // after the x pass: [z][y][ox]
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      int z = iz + dz - 1;
      z = z < 0 ? 0 : (z > Di - 1 ? Di - 1 : z);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        int yy = iy + dy - 1;
        yy = yy < 0 ? 0 : (yy > Hi - 1 ? Hi - 1 : yy);
        const float* row = p + ((long)z * Hi + yy) * Wi;
        float a = row[xa], b = row[ix], cc = row[ix + 1], d = row[xd];
        if (ss) { a = norm_relu1(a, sc, sh); b = norm_relu1(b, sc, sh); cc = norm_relu1(cc, sc, sh); d = norm_relu1(d, sc, sh); }
        r[dz][dy][0] = up2_even(a, b, ix == 0);
        r[dz][dy][1] = up2_odd(b, cc);
        r[dz][dy][2] = up2_even(b, cc, false);
        r[dz][dy][3] = up2_odd(cc, d);
      }
    }
    float u[3][2][4];   // after the y pass: [z][oy parity][ox]
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u[dz][0][k] = up2_even(r[dz][0][k], r[dz][1][k], iy == 0);
        u[dz][1][k] = up2_odd(r[dz][1][k], r[dz][2][k]);
      }
    float* yo = y + (long)n * y_bstride + (((long)c * (2 * Di) + 2 * iz) * Ho + 2 * iy) * Wo + ix * 2;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float ev[4], od[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ev[k] = up2_even(u[0][py][k], u[1][py][k], iz == 0);
        od[k] = up2_odd(u[1][py][k], u[2][py][k]);
      }
      *reinterpret_cast<float4*>(yo + (long)py * Wo) = make_float4(ev[0], ev[1], ev[2], ev[3]);
      *reinterpret_cast<float4*>(yo + ((long)Ho + py) * Wo) = make_float4(od[0], od[1], od[2], od[3]);
    }
  }
}

// Adjoint of the above in ONE pass: input voxel i collects outputs 2i-1, 2i, 2i+1, 2i+2 with weights 0.25, 0.75, 0.75, 0.25
// (i = 0: -, 1, 0.75, 0.25;  i = n-1: 0.25, 0.75, 1, -) along every axis.  One thread = four consecutive input x: per (oz, oy)
// row it reads the ten gradients 8q-1 .. 8q+8 (two float4 + two scalars), reduces along x, then y, then z.  The three-pass
// separable scheme it replaces moves 2.3x the bytes through HBM.  Wi % 4 == 0, 16-byte aligned gradient rows.
__device__ __forceinline__ void up2_adj_weights(int i, int n, float w[4]) {
  w[0] = i > 0 ? 0.25f : 0.f;
  w[1] = i > 0 ? 0.75f : 1.f;
  w[2] = i < n - 1 ? 0.75f : 1.f;
  w[3] = i < n - 1 ? 0.25f : 0.f;
}

// The same with FOUR input voxels along x per thread (Wi % 4 == 0): one aligned 16-byte load and two edge values per input row (27 loads for
// 32 outputs instead of 36 for 16 — the two-voxel kernel streams a 96^3 x 32-channel output at 3.1 TB/s, bound by its load instructions),
// eight 16-byte stores.  Every output is computed by the formulas of the two-voxel kernel: bit-identical.
__global__ __launch_bounds__(256) void upsample2x_fwd_x4_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int Di,
                                                                int Hi, int Wi, long y_bstride, const float* __restrict__ ss) {
  const unsigned wq = (unsigned)Wi >> 2;
  const unsigned total = (unsigned)N * C * Di * Hi * wq;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const unsigned xq = e % wq;
    unsigned t = e / wq;
    const int iy = (int)(t % (unsigned)Hi);
    t /= (unsigned)Hi;
    const int iz = (int)(t % (unsigned)Di);
    t /= (unsigned)Di;
    const int c = (int)(t % (unsigned)C);
    const int n = (int)(t / (unsigned)C);
    const float* p = x + ((long)n * C + c) * Di * Hi * Wi;
    const float sc = ss ? ss[2 * (n * C + c)] : 1.f, sh = ss ? ss[2 * (n * C + c) + 1] : 0.f;
    const int ix = (int)xq * 4;
    const int xa = ix > 0 ? ix - 1 : 0, xd = ix + 4 < Wi ? ix + 4 : Wi - 1;
    float r[3][3][8];   // after the x pass: [z][y][ox]
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      int z = iz + dz - 1;
      z = z < 0 ? 0 : (z > Di - 1 ? Di - 1 : z);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        int yy = iy + dy - 1;
        yy = yy < 0 ? 0 : (yy > Hi - 1 ? Hi - 1 : yy);
        const float* row = p + ((long)z * Hi + yy) * Wi;
        const float4 q = *reinterpret_cast<const float4*>(row + ix);
        float v[6] = {row[xa], q.x, q.y, q.z, q.w, row[xd]};
        if (ss) {
#pragma unroll
          for (int k = 0; k < 6; ++k) v[k] = norm_relu1(v[k], sc, sh);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {     // input ix + k: outputs 2k (from v[k], v[k+1]) and 2k + 1 (from v[k+1], v[k+2])
          r[dz][dy][2 * k] = up2_even(v[k], v[k + 1], ix + k == 0);
          r[dz][dy][2 * k + 1] = up2_odd(v[k + 1], v[k + 2]);
        }
      }
    }
    float u[3][2][8];   // after the y pass: [z][oy parity][ox]
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        u[dz][0][k] = up2_even(r[dz][0][k], r[dz][1][k], iy == 0);
        u[dz][1][k] = up2_odd(r[dz][1][k], r[dz][2][k]);
      }
    float* yo = y + (long)n * y_bstride + (((long)c * (2 * Di) + 2 * iz) * Ho + 2 * iy) * Wo + ix * 2;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float ev[8], od[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        ev[k] = up2_even(u[0][py][k], u[1][py][k], iz == 0);
        od[k] = up2_odd(u[1][py][k], u[2][py][k]);
      }
      float* e0 = yo + (long)py * Wo;
      float* o0 = yo + ((long)Ho + py) * Wo;
      *reinterpret_cast<float4*>(e0) = make_float4(ev[0], ev[1], ev[2], ev[3]);
      *reinterpret_cast<float4*>(e0 + 4) = make_float4(ev[4], ev[5], ev[6], ev[7]);
      *reinterpret_cast<float4*>(o0) = make_float4(od[0], od[1], od[2], od[3]);
      *reinterpret_cast<float4*>(o0 + 4) = make_float4(od[4], od[5], od[6], od[7]);
    }
  }
}

// One thread = four consecutive input x of kUpSeg consecutive input z: the fine planes 2 iz + 1 and 2 iz + 2 it reduces for input
// plane iz are the planes 2 (iz + 1) - 1 and 2 (iz + 1) of the next one, so every fine row is loaded once per thread (32 instead of
// 64 loads per output quad; the remaining 2x re-read across neighbouring iy comes out of L1 / L2).
constexpr int kUpSeg = 8;
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int N, int C, int Di,
                                                             int Hi, int Wi, long gy_bstride) {
  const unsigned wq = (unsigned)Wi >> 2;
  const unsigned nseg = ((unsigned)Di + kUpSeg - 1) / kUpSeg;
  const unsigned total = (unsigned)N * C * nseg * Hi * wq;
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const unsigned xq = e % wq;
    unsigned t = e / wq;
    const int iy = (int)(t % (unsigned)Hi);
    t /= (unsigned)Hi;
    const int seg = (int)(t % nseg);
    t /= nseg;
    const int c = (int)(t % (unsigned)C);
    const int n = (int)(t / (unsigned)C);
    const int ix = (int)xq * 4, ox0 = ix * 2;
    float wy[4], wx[4][4];
    up2_adj_weights(iy, Hi, wy);
#pragma unroll
    for (int k = 0; k < 4; ++k) up2_adj_weights(ix + k, Wi, wx[k]);
    const float* g = gy + (long)n * gy_bstride + (long)c * Do * Ho * Wo;
    // x- and y-reduced values of fine plane oz for the four inputs of this thread (zeros outside the volume)
    auto plane = [&](int oz, float (&tz)[4]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) tz[k] = 0.f;
      if (oz < 0 || oz >= Do) return;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) {
        const int oy = 2 * iy - 1 + dy;
        if (wy[dy] == 0.f) continue;
        const float* row = g + ((long)oz * Ho + oy) * Wo + ox0;
        const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 4);
        const float lo = ox0 > 0 ? row[-1] : 0.f, hi = ox0 + 8 < Wo ? row[8] : 0.f;
        const float v[10] = {lo, v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, hi};
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // input ix+k reads outputs 2k-1 .. 2k+2 relative to ox0, i.e. v[2k .. 2k+3]
          const float xr = wx[k][0] * v[2 * k] + wx[k][1] * v[2 * k + 1] + wx[k][2] * v[2 * k + 2] + wx[k][3] * v[2 * k + 3];
          tz[k] += wy[dy] * xr;
        }
      }
    };
    const int z0 = seg * kUpSeg, z1 = z0 + kUpSeg < Di ? z0 + kUpSeg : Di;
    float pm[4], p0[4], p1[4], p2[4];
    plane(2 * z0 - 1, pm);
    plane(2 * z0, p0);
    for (int iz = z0; iz < z1; ++iz) {
      float wz[4];
      up2_adj_weights(iz, Di, wz);
      plane(2 * iz + 1, p1);
      plane(2 * iz + 2, p2);
      float acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[k] = wz[0] * pm[k];
        acc[k] += wz[1] * p0[k];
        acc[k] += wz[2] * p1[k];
        acc[k] += wz[3] * p2[k];
        pm[k] = p1[k];
        p0[k] = p2[k];
      }
      const long o = ((((long)n * C + c) * Di + iz) * Hi + iy) * wq + xq;
      reinterpret_cast<float4*>(gx)[o] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

// weight with which output index o contributes to input index i along one axis (exact transpose of lin_tap)
__device__ __forceinline__ float lin_w(int o, int i, float rscale, int in_size, int align) {
  const LinTap t = lin_tap(o, rscale, in_size, align);
  return (t.i0 == i ? t.l0 : 0.f) + (t.i1 == i ? t.l1 : 0.f);
}

// conservative [lo, hi] range of output indices that can touch input index i
__device__ __forceinline__ void lin_range(int i, float rscale, int out_size, int align, int& lo, int& hi) {
  if (rscale <= 0.f) { lo = 0; hi = out_size - 1; return; }
  const float inv = 1.0f / rscale;
  if (align) {
    lo = (int)floorf(((float)i - 1.0f) * inv) - 1;
    hi = (int)ceilf(((float)i + 1.0f) * inv) + 1;
  } else {
    lo = (int)floorf(((float)i - 1.0f + 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 1.0f + 0.5f) * inv - 0.5f) + 1;
  }
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}

// Backward of the separable interpolation, one axis per launch (deterministic gather, no atomics):
//   dst[b][r][i][t] = sum_o w(o -> i) * src[b][r][o][t],   r < rows, i < Li, o < Lo, t < inner
// src batches are src_bstride apart (lets the first pass read a channel slice of a concat-gradient
// buffer), dst is dense.  The x pass (inner == 1) runs first on the full-resolution gradient, then y and z
// on tensors that are already 1/scale and 1/scale^2 of it: HBM traffic ~ (1 + 2/s + 2/s^2 + 1/s^3) reads
// of dY instead of one gather per input voxel.  One thread per dst element, t fastest.
__global__ __launch_bounds__(256) void resize_bwd_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, long rows,
                                                              int Li, int Lo, long inner, float rscale, long src_bstride, int align) {
  const long per_b = rows * Li * inner;
  const long total = (long)B * per_b;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / per_b;
    long r = e - b * per_b;
    const long t = r % inner;
    r /= inner;
    const int i = (int)(r % Li);
    const long row = r / Li;
    int lo, hi;
    lin_range(i, rscale, Lo, align, lo, hi);
    const float* p = src + b * src_bstride + (row * Lo) * inner + t;
    float acc = 0.f;
    for (int o = lo; o <= hi; ++o) {
      const float w = lin_w(o, i, rscale, Li, align);
      if (w != 0.f) acc += w * p[(long)o * inner];
    }
    dst[e] = acc;
  }
}

// resize_bwd_axis for the y / z passes (inner % 4 == 0, aligned): four adjacent t per thread share the taps,
// 16-byte loads and stores, 32-bit index arithmetic (host guarantees the quad count fits).
__global__ __launch_bounds__(256) void resize_bwd_axis_t4_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, unsigned rows,
                                                                 int Li, int Lo, unsigned inner4, float rscale, long src_bstride, int align) {
  const unsigned total = (unsigned)B * rows * Li * inner4;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const unsigned t4 = e % inner4;
    unsigned r = e / inner4;
    const int i = (int)(r % (unsigned)Li);
    r /= (unsigned)Li;
    const unsigned row = r % rows;
    const unsigned b = r / rows;
    int lo, hi;
    lin_range(i, rscale, Lo, align, lo, hi);
    const float4* p = reinterpret_cast<const float4*>(src + (long)b * src_bstride + ((long)row * Lo) * inner4 * 4) + t4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight taps' loads in flight, added in tap order (a 96 -> 6 axis gathers up to 32 taps per output: one dependent load per tap made the
    // few-workgroup launches of the deep maps' loss gradients 20-30 us latency chains at the head of the aligner lanes' backward)
    for (int o = lo; o <= hi; o += 8) {
      float4 v[8];
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int oo = o + u <= hi ? o + u : hi;
        w[u] = o + u <= hi ? lin_w(oo, i, rscale, Li, align) : 0.f;
        v[u] = p[(long)oo * inner4];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (w[u] != 0.f) { acc.x += w[u] * v[u].x; acc.y += w[u] * v[u].y; acc.z += w[u] * v[u].z; acc.w += w[u] * v[u].w; }
    }
    reinterpret_cast<float4*>(dst)[e] = acc;
  }
}

// resize_bwd_axis for the x pass (inner == 1, Li % 4 == 0): four adjacent i per thread walk the union of their
// source ranges once, computing each source tap once; one float4 store.
__global__ __launch_bounds__(256) void resize_bwd_axis_i4_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, unsigned rows,
                                                                 int Li, int Lo, float rscale, long src_bstride, int align) {
  const unsigned li4 = (unsigned)Li >> 2;
  const unsigned total = (unsigned)B * rows * li4;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const unsigned iq = e % li4;
    const unsigned r = e / li4;
    const unsigned row = r % rows;
    const unsigned b = r / rows;
    const int i = (int)(iq * 4);
    int lo, hi, lo3, hi3;
    lin_range(i, rscale, Lo, align, lo, hi);
    lin_range(i + 3, rscale, Lo, align, lo3, hi3);  // both ends are monotone in i: the union is [lo, hi3]
    (void)hi; (void)lo3;
    const float* p = src + (long)b * src_bstride + (long)row * Lo;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int o = lo; o <= hi3; ++o) {
      const LinTap t = lin_tap(o, rscale, Li, align);
      const float v = p[o];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float w = (t.i0 == i + k ? t.l0 : 0.f) + (t.i1 == i + k ? t.l1 : 0.f);
        if (w != 0.f) acc[k] += w * v;
      }
    }
    reinterpret_cast<float4*>(dst)[e] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// resize_bwd_axis for the x pass (inner == 1) of LARGE upscales (the 6/12/24 -> 96 maps of the deep-supervision losses, 16 classes:
// 56 MB of dY): the per-thread walk of resize_bwd_axis_i4_kernel reads dY four bytes at a time with a 16-output stride between
// lanes (147 us for 56 MB).  Here a workgroup stages kRbxRows complete rows of dY in LDS with coalesced 16-byte loads and every
// thread then gathers its (row, i) sums from LDS — same taps, same summation order over o as the other variants.
// Lo % 4 == 0, 16-byte aligned rows; LDS = kRbxRows * (Lo + 4) floats; grid ceil(B * rows / kRbxRows).
constexpr int kRbxRows = 32;
__global__ __launch_bounds__(256) void resize_bwd_axis_x_lds_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, long rows,
                                                                    int Li, int Lo, float rscale, long src_bstride, int align) {
  ICL_DYN_LDS(float, tile);
  const int pitch = Lo + 4, lo4 = Lo >> 2;
  const long total_rows = (long)B * rows;
  const long r0 = (long)blockIdx.x * kRbxRows;
  for (int it = threadIdx.x; it < kRbxRows * lo4; it += 256) {
    const int row = it / lo4, q = it - row * lo4;
    const long gr = r0 + row;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gr < total_rows) {
      const long b = gr / rows, rr = gr - b * rows;
      v = *reinterpret_cast<const float4*>(src + b * src_bstride + rr * Lo + q * 4);
    }
    *reinterpret_cast<float4*>(tile + row * pitch + q * 4) = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kRbxRows * Li; e += 256) {
    const int row = e / Li, i = e - row * Li;
    const long gr = r0 + row;
    if (gr >= total_rows) break;
    int lo, hi;
    lin_range(i, rscale, Lo, align, lo, hi);
    const float* p = tile + row * pitch;
    float acc = 0.f;
    for (int o = lo; o <= hi; ++o) {
      const float w = lin_w(o, i, rscale, Li, align);
      if (w != 0.f) acc += w * p[o];
    }
    dst[gr * Li + i] = acc;
  }
}

// dst[r*dst_stride + i] = src[r*src_stride + i], i < row_elems
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows,
                                                        long row_elems, long src_stride, long dst_stride) {
  const long total = rows * row_elems;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / row_elems, i = e - r * row_elems;
    dst[r * dst_stride + i] = src[r * src_stride + i];
  }
}

// dst[n][c][:] = relu(fma(src[n][c][:], scale, shift)) for the skip half of a concat buffer whose source is a deferred normalisation:
// rows = N * C channel rows of S floats (S % 4 == 0), the destination sample stride differs from the source's (the concat buffer).
__global__ __launch_bounds__(256) void copy_rows_norm_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, long S,
                                                             long dst_bstride, const float* __restrict__ ss) {
  const long S4 = S >> 2, total = (long)N * C * S4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long q = e % S4, r = e / S4;
    const int c = (int)(r % C), n = (int)(r / C);
    const float4 v = *reinterpret_cast<const float4*>(src + r * S + (q << 2));
    *reinterpret_cast<float4*>(dst + (long)n * dst_bstride + (long)c * S + (q << 2)) = norm_relu4(v, ss[2 * r], ss[2 * r + 1]);
  }
}

// out = [a ; b] (na + nb floats, multiples of 4); a null half is written as zeros: the gradient of a batch split of which only one part was
// used (x[:k], x[k:] in the ICL forward) in one launch instead of a zero fill, a copy and a concatenation.
__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, long na, const float* __restrict__ b, long nb,
                                                      float* __restrict__ out) {
  const long nq = (na + nb) >> 2, aq = na >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
    const float* src = q < aq ? a : b;
    const long i = q < aq ? q : q - aq;
    reinterpret_cast<float4*>(out)[q] = src ? reinterpret_cast<const float4*>(src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ---- depth-to-space / space-to-depth for ConvTranspose3d(kernel 2, stride 2) written as a GEMM (MONAI UnetrUpBlock.transp_conv):
// the GEMM produces tokens yt[b][(z,y,x)][co*8 + i*4 + j*2 + k]; the volume is out[b][co][2z+i][2y+j][2x+k].
// One workgroup moves one token row (b, z, y, all x): it reads W contiguous token vectors (cout*8 floats each) into LDS and
// writes, for every (co, i, j), one contiguous run of 2W floats — both sides of the copy are full-line accesses (the generic
// strided permute reaches 1.7 TB/s on the 96^3 layer).  LDS row pitch cout*8 + 2 keeps the transposed reads conflict-free.
// out_bstride: batch stride of `out` in floats (lets the caller write the channel slice of a concat buffer).
// grid (H, D, N), block 256, LDS = W * (cout*8 + 2) floats.
__global__ __launch_bounds__(256) void depth_to_space2_kernel(const float* __restrict__ yt, float* __restrict__ out, int D, int H, int W,
                                                              int cout, long out_bstride) {
  ICL_DYN_LDS(float, tile);
  const int y = blockIdx.x, z = blockIdx.y, b = blockIdx.z;
  const int C8 = cout * 8, pitch = C8 + 2;
  const float* src = yt + (((long)b * D + z) * H + y) * (long)W * C8;
  for (int it = threadIdx.x; it < W * C8; it += blockDim.x) tile[(it / C8) * pitch + it % C8] = src[it];
  __syncthreads();
  const int run = 2 * W;
  const long Ho = 2L * H, Wo = 2L * W, So = 2L * D * Ho * Wo;
  for (int it = threadIdx.x; it < cout * 4 * run; it += blockDim.x) {
    const int X = it % run;
    const int r = it / run;
    const int j = r & 1, i = (r >> 1) & 1, co = r >> 2;
    out[(long)b * out_bstride + (long)co * So + ((2L * z + i) * Ho + 2 * y + j) * Wo + X] =
        tile[(X >> 1) * pitch + co * 8 + i * 4 + j * 2 + (X & 1)];
  }
}

// exact inverse (backward of the above): g has the layout of `out` (batch stride g_bstride), gt the layout of yt.
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float* __restrict__ g, float* __restrict__ gt, int D, int H, int W,
                                                              int cout, long g_bstride) {
  ICL_DYN_LDS(float, tile);
  const int y = blockIdx.x, z = blockIdx.y, b = blockIdx.z;
  const int C8 = cout * 8, pitch = C8 + 2;
  const int run = 2 * W;
  const long Ho = 2L * H, Wo = 2L * W, So = 2L * D * Ho * Wo;
  for (int it = threadIdx.x; it < cout * 4 * run; it += blockDim.x) {
    const int X = it % run;
    const int r = it / run;
    const int j = r & 1, i = (r >> 1) & 1, co = r >> 2;
    tile[(X >> 1) * pitch + co * 8 + i * 4 + j * 2 + (X & 1)] =
        g[(long)b * g_bstride + (long)co * So + ((2L * z + i) * Ho + 2 * y + j) * Wo + X];
  }
  __syncthreads();
  float* dst = gt + (((long)b * D + z) * H + y) * (long)W * C8;
  for (int it = threadIdx.x; it < W * C8; it += blockDim.x) dst[it] = tile[(it / C8) * pitch + it % C8];
}

}  // namespace icl
