// misc.h — depthwise 3x3x3 convolution (SeparableConv3d.depthwise, /root/reference/code/networks/
// unet_3D_icl.py:320-323) and counter-based dropout (nn.Dropout(p=0.3), unet_3D_icl.py:67-68).
// Both HBM-bound elementwise/stencil kernels, x fastest for coalescing.
#pragma once

namespace icl {

// y[n,c,z,y,x] = sum_tap w[c][tap] * x[n,c,z+dz-1,y+dy-1,x+dx-1]; flip=1 uses w[c][26-tap] (dgrad).
__global__ __launch_bounds__(256) void dwconv3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                          long NC, int C, int D, int H, int W, int flip) {
  const long S = (long)D * H * W;
  const long total = NC * S;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(o % W);
    long t = o / W;
    const int oy = (int)(t % H);
    t /= H;
    const int oz = (int)(t % D);
    const long nc = t / D;
    const int c = (int)(nc % C);
    const float* xp = x + nc * S;
    const float* wc = w + c * 27;
    float acc = 0.f;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      const int z = oz + dz - 1;
      if (z < 0 || z >= D) continue;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = oy + dy - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = ox + dx - 1;
          if (xx < 0 || xx >= W) continue;
          const int tap = (dz * 3 + dy) * 3 + dx;
          acc += wc[flip ? 26 - tap : tap] * xp[((long)z * H + yy) * W + xx];
        }
      }
    }
    y[o] = acc;
  }
}

// slab[chunk][c][tap] = sum over a chunk of (n, voxels) of gy * shifted x.  grid (chunks, C): the 27 sums are reduced inside each
// wave with shuffles and across the four waves through LDS (one barrier); dwconv3_wgrad_reduce_kernel adds the chunks in a fixed
// order (no atomics: reproducible).
__global__ __launch_bounds__(256) void dwconv3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ slab,
                                                            int N, int C, int D, int H, int W, long chunk) {
  const int c = blockIdx.y;
  const long S = (long)D * H * W;
  const long total = (long)N * S;
  const long lo = (long)blockIdx.x * chunk;
  const long hi = lo + chunk < total ? lo + chunk : total;
  float acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.f;
  for (long e = lo + threadIdx.x; e < hi; e += 256) {
    const long n = e / S;
    const long v = e - n * S;
    const int ox = (int)(v % W);
    const long t = v / W;
    const int oy = (int)(t % H), oz = (int)(t / H);
    const float g = gy[(n * C + c) * S + v];
    const float* xp = x + (n * C + c) * S;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int z = oz + dz - 1, yy = oy + dy - 1, xx = ox + dx - 1;
          if (z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W)
            acc[(dz * 3 + dy) * 3 + dx] += g * xp[((long)z * H + yy) * W + xx];
        }
      }
    }
  }
  __shared__ float red[4][28];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 27; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[wid][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 27)
    slab[((long)blockIdx.x * C + c) * 27 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Block 256 = 16 consecutive elements x 16 chunk lanes: lane l adds chunks l, l + 16, ... (independent loads, eight in flight), the 16
// partial sums of an element are combined through LDS in lane order — a fixed order, no atomics.  (Until round 6 one thread per element
// walked all chunks: 53 us for 64 chunks on SwinUNETR's 48^3 maps.)  grid ceil(E / 16).
__global__ __launch_bounds__(256) void dwconv3_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ gw, int E, int chunks) {
  __shared__ float red[16][17];
  const int le = threadIdx.x & 15, ls = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + le;
  float v = 0.f;
  if (e < E) {
#pragma unroll 8
    for (int k = ls; k < chunks; k += 16) v += slab[(long)k * E + e];
  }
  red[ls][le] = v;
  __syncthreads();
  if (ls != 0 || e >= E) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) v += red[k][le];
  gw[e] = v;
}

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

// The element-wise dropout mask of dropout_kernel (below) for kernels that apply it on the fly: keep(k) = hash(seed, k) >= thresh, k = the flat
// index of the element in the dropped tensor.  mode 0: off; 1: the kernel's INPUT x is the dropped tensor (forward of dropout -> conv, and the
// weight gradient's x operand); 2: its OUTPUT is (the input gradient: d(dropout) applied to what the convolution's backward produces).
struct DropSpec {
  unsigned seed, thresh;
  float scale;
  const unsigned* seed_dev;
  int mode;
  const float* ss;      // != nullptr: the kernel's x operand is a RAW convolution output; relu(fma(x, scale, shift)) per (sample, channel) is
                        // applied on load, in front of the mask (deferred InstanceNorm + ReLU, norm.h norm_finalize_stats_kernel)
};
__device__ __forceinline__ unsigned drop_seed(const DropSpec& d) { return d.seed_dev ? mix32(d.seed ^ mix32(*d.seed_dev + 0x632BE5ABu)) : d.seed; }
__device__ __forceinline__ float drop_apply(float v, unsigned seed, unsigned thresh, float scale, long k) {
  const unsigned h = mix32((unsigned)k * 0x9E3779B1u + seed) ^ mix32((unsigned)(k >> 32) + seed * 0x85EBCA77u);
  return h >= thresh ? v * scale : 0.f;
}
__device__ __forceinline__ float4 drop_apply4(float4 v, unsigned seed, unsigned thresh, float scale, long k) {
  return make_float4(drop_apply(v.x, seed, thresh, scale, k), drop_apply(v.y, seed, thresh, scale, k + 1), drop_apply(v.z, seed, thresh, scale, k + 2),
                     drop_apply(v.w, seed, thresh, scale, k + 3));
}

// ---- 1x1x1 convolution with a handful of channels on a small volume (the aligner's SeparableConv3d.pointwise h -> h and attn_convs1
// h -> 1 on <= 4 x 24^3 maps, unet_3D_icl.py:196,327, and their input gradients): y[b][o][v] = bias[o] + sum_i w(o, i) * x[b][i][v],
// w(o, i) = w[o * w_ostride + i * w_istride] (the input gradient is the same kernel with the strides swapped).  One thread per
// (b, o, four consecutive voxels), 16-byte loads and stores; the implicit-GEMM kernel pads such a layer to 16 output channels
// and a halo tile and takes 10-14 us for it.
__global__ __launch_bounds__(256) void conv1x1_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int N, int CI, int CO,
                                                            long S, int wos, int wis) {
  const long S4 = S >> 2, total = (long)N * CO * S4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long v4 = e % S4, r = e / S4;
    const int o = (int)(r % CO);
    const long b = r / CO;
    const float b0 = bias ? bias[o] : 0.f;
    float4 acc = make_float4(b0, b0, b0, b0);
    const float* xp = x + b * CI * S + (v4 << 2);
    const float* wp = w + (long)o * wos;
    for (int i = 0; i < CI; ++i) {
      const float4 xv = *reinterpret_cast<const float4*>(xp + (long)i * S);
      const float wv = wp[(long)i * wis];
      acc.x = fmaf(wv, xv.x, acc.x);
      acc.y = fmaf(wv, xv.y, acc.y);
      acc.z = fmaf(wv, xv.z, acc.z);
      acc.w = fmaf(wv, xv.w, acc.w);
    }
    *reinterpret_cast<float4*>(y + (b * CO + o) * S + (v4 << 2)) = acc;
  }
}

// ---- the same 1x1x1 convolution for BIG volumes (the `final` 16 -> num_classes convolution on 96^3 voxels, unet_3D_icl.py:65,117,
// and its input gradient): one thread owns four consecutive voxels and OB output channels, so every input value is loaded once
// per OB outputs (the kernel above re-reads the input per output channel) and the weights are wave-uniform scalar loads.
// HBM-bound: 4 (Cin + Cout) bytes per voxel.  grid (voxel-quad blocks, ceil(CO / OB)).
template <int OB>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, int N, int CI, int CO,
                                                             long S, int wos, int wis, DropSpec dr) {
  const long S4 = S >> 2, total = (long)N * S4;
  const int o0 = blockIdx.y * OB;
  const unsigned dseed = dr.mode ? drop_seed(dr) : 0u;
  float wb[OB];
#pragma unroll
  for (int o = 0; o < OB; ++o) wb[o] = (bias && o0 + o < CO) ? bias[o0 + o] : 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long v4 = e % S4, b = e / S4;
    float4 acc[OB];
#pragma unroll
    for (int o = 0; o < OB; ++o) acc[o] = make_float4(wb[o], wb[o], wb[o], wb[o]);
    const float* xp = x + b * CI * S + (v4 << 2);
    for (int i = 0; i < CI; ++i) {
      float4 xv = *reinterpret_cast<const float4*>(xp + (long)i * S);
      if (dr.ss) xv = norm_relu4(xv, dr.ss[2 * (b * CI + i)], dr.ss[2 * (b * CI + i) + 1]);
      if (dr.mode == 1) xv = drop_apply4(xv, dseed, dr.thresh, dr.scale, (b * CI + i) * S + (v4 << 2));      // dropout -> conv: the mask on load
#pragma unroll
      for (int o = 0; o < OB; ++o) {
        const float wv = o0 + o < CO ? w[(long)(o0 + o) * wos + (long)i * wis] : 0.f;
        acc[o].x = fmaf(wv, xv.x, acc[o].x);
        acc[o].y = fmaf(wv, xv.y, acc[o].y);
        acc[o].z = fmaf(wv, xv.z, acc[o].z);
        acc[o].w = fmaf(wv, xv.w, acc[o].w);
      }
    }
#pragma unroll
    for (int o = 0; o < OB; ++o)
      if (o0 + o < CO) {
        if (dr.mode == 2) acc[o] = drop_apply4(acc[o], dseed, dr.thresh, dr.scale, (b * CO + o0 + o) * S + (v4 << 2));      // input gradient: the mask on store
        *reinterpret_cast<float4*>(y + (b * CO + o0 + o) * S + (v4 << 2)) = acc[o];
      }
  }
}

// ---- out[j] = sum_i w[j][i] * *in[i] for up to 16 device scalars in and 16 out: every "+", "/ 3", "10 *" between the loss terms of
// a step (utils/losses.py, trainer :105-112) in ONE launch — as 0-dim torch arithmetic they were ~20 one-element kernels forward and
// ~10 backward.  The backward is the same kernel with the transposed weights (the inputs then are the elements of one gradient vector).
constexpr int kScalarCombine = 16;
struct ScalarCombine {
  const float* in[kScalarCombine];
  float w[kScalarCombine][kScalarCombine];      // [out][in]
  int n_in, n_out;
};
__global__ __launch_bounds__(64) void scalar_combine_kernel(ScalarCombine s, float* __restrict__ out) {
  const int j = threadIdx.x;
  if (j >= s.n_out) return;
  float acc = 0.f;
  for (int i = 0; i < s.n_in; ++i) acc += s.w[j][i] * *s.in[i];      // fixed order
  out[j] = acc;
}

// ---- column sums of up to kColsumMulti small row-major matrices in ONE launch (grid.y = matrix): out_i[c] = sum_r g_i[r][c].
// The bias gradients of the aligner's Linear layers (a few to a few hundred rows each): the trainer collects them during backward
// and reduces them together instead of one tiny reduction launch per layer (38 per U-Net step).
constexpr int kColsumMulti = 48;
struct ColsumMulti {
  const float* g[kColsumMulti];
  float* out[kColsumMulti];
  int rows[kColsumMulti], cols[kColsumMulti];
};

// Block 256 = 16 consecutive columns x 16 row lanes (every row is read as a 64-byte segment, 16 rows in flight per block); the 16
// partial sums of a column are combined through LDS in lane order — a fixed summation order.  grid (ceil(max cols / 16), matrices).
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumMulti m) {
  __shared__ float red[16][17];
  const int i = blockIdx.y;
  const int R = m.rows[i], C = m.cols[i];
  const float* __restrict__ g = m.g[i];
  const int lc = threadIdx.x & 15, lr = threadIdx.x >> 4;
  for (int c0 = blockIdx.x * 16; c0 < C; c0 += gridDim.x * 16) {      // block-uniform loop
    const int c = c0 + lc;
    float acc = 0.f;
    if (c < C) {
#pragma unroll 4
      for (int r = lr; r < R; r += 16) acc += g[(long)r * C + c];
    }
    __syncthreads();
    red[lr][lc] = acc;
    __syncthreads();
    if (lr == 0 && c < C) {
#pragma unroll
      for (int k = 1; k < 16; ++k) acc += red[k][lc];
      m.out[i][c] = acc;
    }
  }
}

// ---- row gather: out[b][m][:] = idx[m] >= 0 ? src[b][idx[m]][:] : 0   (rows of C floats, C % 4 == 0).
// The Swin blocks move tokens between the volume order and the (rolled, zero-padded) window order
// (zero pad -> torch.roll -> window_partition and window_reverse -> roll back -> crop, networks/swinunetr_icl.py:825-866): both
// directions, and both of their gradients, are ONE gather with a precomputed index (every token has exactly one window slot;
// padded slots carry -1) instead of three full-tensor copies each.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ out,
                                                          long B, long S, long M, int C4) {
  const long total = B * M * C4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    const long r = e / C4;
    const long m = r % M, b = r / M;
    const int s = idx[m];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0) v = reinterpret_cast<const float4*>(src)[(b * S + s) * C4 + c];
    reinterpret_cast<float4*>(out)[e] = v;
  }
}

// out[b][m][:] = src[b][idx2[m][0]][:] + src[b][idx2[m][1]][:]  (either index may be -1 = absent).  Gradient of a row gather whose
// index lists some source rows twice and some never: the reference's PatchMerging concatenates eight strided slices of which two
// are repeated and two missing (networks/swinunetr_icl.py:953-961), so a token's gradient is the sum of 0, 1 or 2 gathered rows.
__global__ __launch_bounds__(256) void gather_rows_sum2_kernel(const float* __restrict__ src, const int* __restrict__ idx2,
                                                               float* __restrict__ out, long B, long S, long M, int C4) {
  const long total = B * M * C4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C4);
    const long r = e / C4;
    const long m = r % M, b = r / M;
    const int s0 = idx2[2 * m], s1 = idx2[2 * m + 1];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s0 >= 0) v = reinterpret_cast<const float4*>(src)[(b * S + s0) * C4 + c];
    if (s1 >= 0) {
      const float4 u = reinterpret_cast<const float4*>(src)[(b * S + s1) * C4 + c];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    reinterpret_cast<float4*>(out)[e] = v;
  }
}

// ---- im2col / col2im for 3^3 convolutions on TINY volumes (<= 6^3: the 384/768-channel bottleneck blocks of SwinUNETR and the
// 256-channel centre of the U-Net).  There the convolution is a skinny GEMM that streams up to 64 MB of weights for 27..432
// output voxels, so it runs as  colsT [N*S, Cin*27] x W[Cout, Cin*27]^T  on the library GEMM; these two kernels only move
// the (small) activations.  colsT[b*S + v][ci*27 + tap] = x[b][ci][v + offset(tap)] (zero outside the volume).
__global__ __launch_bounds__(256) void im2col3_kernel(const float* __restrict__ x, float* __restrict__ cols, int N, int C, int D, int H,
                                                      int W) {
  const long S = (long)D * H * W, K = (long)C * 27, total = (long)N * S * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % K);
    const long r = e / K;
    const int tap = k % 27, ci = k / 27;
    const long v = r % S;
    const int b = (int)(r / S);
    const int ox = (int)(v % W), oy = (int)((v / W) % H), oz = (int)(v / ((long)W * H));
    const int z = oz + tap / 9 - 1, yy = oy + (tap / 3) % 3 - 1, xx = ox + tap % 3 - 1;
    float val = 0.f;
    if (z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W) val = x[((long)b * C + ci) * S + ((long)z * H + yy) * W + xx];
    cols[e] = val;
  }
}

// Channel-major variant for the FIRST convolution (Cin = 1..2 on a big volume): planes[b][ci*27 + tap][v] = x[b][ci][v + offset(tap)].
// With one input channel the implicit-GEMM weight gradient pads Cin to 16 and spends 300 us on 0.8 GFLOP; as 27 shifted planes
// the same gradient is the HBM-bound 1x1x1 reduction  dW[co][tap] = sum_v gy[co][v] * planes[tap][v]  (conv1x1_wgrad_kernel).
// One thread per four consecutive v (W % 4 == 0): 16-byte stores.
__global__ __launch_bounds__(256) void im2col3_planes_kernel(const float* __restrict__ x, float* __restrict__ planes, int N, int C, int D,
                                                             int H, int W) {
  const long S = (long)D * H * W, S4 = S >> 2;
  const long total = (long)N * C * 27 * S4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long v = (e % S4) << 2;
    const long r = e / S4;
    const int tap = (int)(r % 27);
    const long bc = r / 27;
    const int ox = (int)(v % W), oy = (int)((v / W) % H), oz = (int)(v / ((long)W * H));
    const int z = oz + tap / 9 - 1, yy = oy + (tap / 3) % 3 - 1, dx = tap % 3 - 1;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (z >= 0 && z < D && yy >= 0 && yy < H) {
      const float* row = x + bc * S + ((long)z * H + yy) * W;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int xx = ox + k + dx;
        if (xx >= 0 && xx < W) o[k] = row[xx];
      }
    }
    *reinterpret_cast<float4*>(planes + r * S + v) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// dx[b][ci][u] = sum_tap g[b*S + (u - offset(tap))][ci*27 + tap]   (exact transpose of im2col3)
__global__ __launch_bounds__(256) void col2im3_kernel(const float* __restrict__ g, float* __restrict__ dx, int N, int C, int D, int H, int W) {
  const long S = (long)D * H * W, K = (long)C * 27, total = (long)N * C * S;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long u = e % S;
    const int ci = (int)((e / S) % C);
    const int b = (int)(e / (S * C));
    const int ux = (int)(u % W), uy = (int)((u / W) % H), uz = (int)(u / ((long)W * H));
    float acc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int z = uz - (tap / 9 - 1), yy = uy - ((tap / 3) % 3 - 1), xx = ux - (tap % 3 - 1);
      if (z >= 0 && z < D && yy >= 0 && yy < H && xx >= 0 && xx < W)
        acc += g[((long)b * S + ((long)z * H + yy) * W + xx) * K + (long)ci * 27 + tap];
    }
    dx[e] = acc;
  }
}

// y = keep ? x*scale : 0 with keep(i) = hash(seed, i) >= thresh; the same (seed) regenerates the mask in backward.
// seed_dev (optional): a device-resident step counter HASHED into the seed, so replays of a captured hipGraph draw new masks
// (mixing it in linearly would make the mask of step t the mask of step 0 shifted by t elements).
// group > 1: one decision per `group` consecutive elements — per-sample stochastic depth (DropPath: the whole residual branch
// of a sample is kept, scaled by 1/(1-p), or dropped).
// res (optional): y = res + dropped(x) — the residual form  x + DropPath(branch)  of the transformer blocks in one pass.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
                                                      long n, unsigned seed, unsigned thresh, float scale,
                                                      const unsigned* __restrict__ seed_dev, long group) {
  if (seed_dev) seed = mix32(seed ^ mix32(*seed_dev + 0x632BE5ABu));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long k = group > 1 ? i / group : i;
    const float v = drop_apply(x[i], seed, thresh, scale, k);
    y[i] = res ? res[i] + v : v;
  }
}

}  // namespace icl
