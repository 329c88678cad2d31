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
// so weight staging is a coalesced 16-B row copy and the MFMA B operand is bank-conflict free.
//
// forward / dgrad kernel (one workgroup = one TZxTYxTX output tile x NB output channels x a Cin split):
//   GEMM view  M = voxels (16 consecutive tile voxels per MFMA), N = Cout (16 per MFMA),
//              K = taps x Cin, walked tap-major in steps of 4 input channels.
//   A[v][k]  = Xs[cin plane k][voxel v + tap offset]   (input halo tile staged in LDS, zero filled)
//   B[k][n]  = Ws[tap*KC + k][n]
//   LDS plane stride PS == 16 (mod 32): lanes 0-15 (voxels) and 16-31 (next cin plane) hit disjoint banks.
//   Staging: with W % 4 == 0 every LDS row holds the 16-B aligned span [x0-4, x0+TX+4) so global loads and
//   LDS stores are 16 B per lane; each thread's (global offset, LDS offset) pairs are computed once and reused
//   for every Cin chunk, and the loads of chunk i+1 are issued into registers before the MFMAs of chunk i.
//   ksplit > 1 spreads the Cin chunks of one output tile over several workgroups; each split writes its partial tile
//   to its own slab and splitk_reduce_kernel adds them in a fixed order (no atomics: measured 1.6x faster than fp32
//   atomics on 64->64 @24^3, and bitwise reproducible): the 6^3..12^3 layers have too few tiles to fill 256 CUs otherwise.
// wgrad kernel (one workgroup = 16 output channels x 16 input channels x all taps, loops over tiles):
//   GEMM view  M = Cout (16), N = Cin (16) per tap, K = voxels in steps of 4 consecutive x.
//   A[co][v] = Gs[co][v] (dY tile, zero where outside the volume), B[v][ci] = Xs[ci][v + tap offset]
//   taps are dealt round-robin to the waves; partial dW of each (split, batch) workgroup goes to its own packed slab, reduced by reduce_unpack_wgrad_kernel.
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
  int KC;      // input channels per LDS chunk (multiple of 4, divides CinP)
  int PS;      // LDS plane stride in floats
  int PXL;     // LDS row pitch in floats
  int HX;      // x halo staged on each side of a row (4 for vector staging of a 3^3 conv, else PAD)
  int CPP;     // channels covered by one staging pass of the workgroup
  int ksplit;  // Cin-chunk split across workgroups
  long x_bstride, y_bstride;  // batch strides in elements (channel stride is D*H*W)
  float* slab;  // ksplit > 1: partial outputs [ksplit][batch][Cout][D*H*W], summed (+bias) by splitk_reduce_kernel
  int remap;    // 1: XCD-aware workgroup order (gridDim.x % 8 == 0), see xcd_chunked()
};

// Workgroups are handed to the eight XCDs round-robin (linear id % 8) and every XCD has its own L2: with tiles numbered in the
// launch order, spatial neighbours — which share their halo — sit on different L2s.  This maps the physical blockIdx.x to a
// logical index so that each XCD owns one contiguous eighth of the index range (valid when gridDim.x % 8 == 0, where
// blockIdx.x % 8 is the XCD for every blockIdx.y/z).
__device__ __forceinline__ int xcd_chunked(int bx, int nb) { return (bx & 7) * (nb >> 3) + (bx >> 3); }

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

// Both packings in one launch (forward layout into wp0, dgrad layout into wp1): a training step needs both of every weight.
__global__ __launch_bounds__(256) void pack_weights_both_kernel(const float* __restrict__ w, float* __restrict__ wp0, float* __restrict__ wp1,
                                                                int Cout, int Cin, int T, int KP0, int NP0, int KP1, int NP1) {
  const long t0 = (long)T * KP0 * NP0, t1 = (long)T * KP1 * NP1;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < t0 + t1; e += (long)gridDim.x * blockDim.x) {
    const bool dg = e >= t0;
    const long f = dg ? e - t0 : e;
    const int NP = dg ? NP1 : NP0, KP = dg ? KP1 : KP0;
    const int n = (int)(f % NP);
    const long t = f / NP;
    const int k = (int)(t % KP);
    const int tap = (int)(t / KP);
    const int K = dg ? Cout : Cin, N = dg ? Cin : Cout;
    float v = 0.f;
    if (k < K && n < N) v = dg ? w[((long)k * Cin + n) * T + (T - 1 - tap)] : w[((long)n * Cin + k) * T + tap];
    (dg ? wp1 : wp0)[f] = v;
  }
}

// Both packings of up to kPackMulti weights in ONE launch (grid.y = weight): the weights only change in the optimiser step, so a
// training step packs every convolution weight once, up front, instead of one small launch per convolution call.
constexpr int kPackMulti = 32;
struct PackMulti {
  const float* w[kPackMulti];
  float* wp0[kPackMulti];
  float* wp1[kPackMulti];
  int cout[kPackMulti], cin[kPackMulti], taps[kPackMulti];
};

// Round 6 (second half): through LDS tiles of 16 couts x 16 cins x T taps.  The element-per-thread form above gathers its reads with a
// stride of Cin * T floats — 418 us for SwinUNETR's weights (16 M packed elements for 768 -> 384 alone), 55-68 us for the U-Net's, one
// launch at the head of every step with nothing beside it.  Here a tile's rows are read as contiguous runs of 16 T floats and both
// layouts are written in 64-byte segments (16 consecutive couts of wp0, 16 consecutive cins of wp1).  Same values, zero padding included.
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(PackMulti m) {
  const int i = blockIdx.y;
  const int Cout = m.cout[i], Cin = m.cin[i], T = m.taps[i];
  const int KP0 = (Cin + 3) / 4 * 4, NP0 = (Cout + 15) / 16 * 16, KP1 = (Cout + 3) / 4 * 4, NP1 = (Cin + 15) / 16 * 16;
  const float* __restrict__ w = m.w[i];
  float* __restrict__ wp0 = m.wp0[i];
  float* __restrict__ wp1 = m.wp1[i];
  constexpr int RS = 16 * 27 + 1;                      // row stride of the tile in LDS (odd: the transposed reads below are conflict-free)
  __shared__ float tile[16 * RS];
  const int ntn = NP0 / 16, ntk = NP1 / 16, run = 16 * T;
  const int tid = threadIdx.x, lo = tid & 15, hi = tid >> 4;
  for (int t = blockIdx.x; t < ntn * ntk; t += gridDim.x) {
    const int n0 = (t / ntk) * 16, k0 = (t % ntk) * 16;
    const int kv = Cin - k0 < 16 ? Cin - k0 : 16;      // cins of this tile that exist
    __syncthreads();                                    // the previous tile has been written out
    for (int q = tid; q < 16 * run; q += 256) {
      const int n = q / run, r = q % run;               // r = k * T + tap inside the row's run
      float v = 0.f;
      if (n0 + n < Cout && r < kv * T) v = w[((long)(n0 + n) * Cin + k0) * T + r];
      tile[n * RS + r] = v;
    }
    __syncthreads();
    // wp0[tap][k][n]: lanes lo = n (64 bytes), hi = k
    if (k0 + hi < KP0)
      for (int tap = 0; tap < T; ++tap) wp0[((long)tap * KP0 + k0 + hi) * NP0 + n0 + lo] = tile[lo * RS + hi * T + tap];
    // wp1[T - 1 - tap][k' = cout][n' = cin]: lanes lo = cin (64 bytes), hi = cout
    if (n0 + hi < KP1)
      for (int tap = 0; tap < T; ++tap) wp1[((long)(T - 1 - tap) * KP1 + n0 + hi) * NP1 + k0 + lo] = tile[hi * RS + lo * T + tap];
  }
}

// gW[co][ci][tap] = sum over slabs s of gWp[s][tap][ci][co]  (fixed summation order -> reproducible).
// A workgroup owns 64 consecutive packed elements; its 4 waves take slabs s = wave, wave+4, ... so every slab
// read is a coalesced 256-byte row, then the four partial sums are combined through LDS in wave order.
__global__ __launch_bounds__(256) void reduce_unpack_wgrad_kernel(const float* __restrict__ gwp, float* __restrict__ gw, int Cout, int Cin,
                                                                  int T, int CinP, int CoutP, int nslabs) {
  const long pe = (long)T * CinP * CoutP;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __shared__ float part[4][64];
  for (long base = (long)blockIdx.x * 64; base < pe; base += (long)gridDim.x * 64) {
    const long idx = base + lane;
    float acc = 0.f;
    if (idx < pe) {
      // (eight loads in flight per lane, added in slab order: the sum is the chain's, the latency is not)
#pragma unroll 8
      for (int s = wid; s < nslabs; s += 4) acc += gwp[s * pe + idx];
    }
    __syncthreads();
    part[wid][lane] = acc;
    __syncthreads();
    if (wid == 0 && idx < pe) {
      const float v = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
      const int co = (int)(idx % CoutP);
      const long t = idx / CoutP;
      const int ci = (int)(t % CinP);
      const int tap = (int)(t / CinP);
      if (co < Cout && ci < Cin) gw[((long)co * Cin + ci) * T + tap] = v;
    }
  }
}

// The same reduction for MANY slabs of FEW elements (the 16-channel layers on 96^3: 512-1024 slabs of 6,912 packed elements; with
// the kernel above 108 workgroups each walk 128-256 slabs per wave, 35 us).  Block 256 = 16 consecutive packed elements x 16 slab
// lanes: slab lane l adds slabs l, l+16, ... (64-byte row segments, independent loads), the 16 partial sums are combined through
// LDS in lane order (fixed order -> reproducible).  grid ceil(pe / 16).
__global__ __launch_bounds__(256) void reduce_unpack_wgrad_tall_kernel(const float* __restrict__ gwp, float* __restrict__ gw, int Cout,
                                                                       int Cin, int T, int CinP, int CoutP, int nslabs) {
  __shared__ float part[16][17];
  const long pe = (long)T * CinP * CoutP;
  const int le = threadIdx.x & 15, ls = threadIdx.x >> 4;
  const long idx = (long)blockIdx.x * 16 + le;
  float acc = 0.f;
  if (idx < pe) {
#pragma unroll 8
    for (int s = ls; s < nslabs; s += 16) acc += gwp[s * pe + idx];
  }
  part[ls][le] = acc;
  __syncthreads();
  if (ls != 0 || idx >= pe) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) acc += part[k][le];
  const int co = (int)(idx % CoutP);
  const long t = idx / CoutP;
  const int ci = (int)(t % CinP);
  const int tap = (int)(t / CinP);
  if (co < Cout && ci < Cin) gw[((long)co * Cin + ci) * T + tap] = acc;
}

// Both reductions above for MANY weight gradients in one launch (round 6): inside a step scope the slab sums of all convolution
// weight gradients of a backward pass are leaves — only the optimiser reads them — so they are queued (ops.DeferredWgradReduce) and
// summed by ONE launch when the pass ends instead of one launch behind every weight-gradient kernel (23 per U-Net step).  Job j takes
// the element-per-slab-lane form of the kernel its stand-alone launch would have taken (`tall`), so every sum keeps its order:
// bit-identical results.  grid (blocks, jobs), grid-stride over the job's packed elements.
constexpr int kReduceMulti = 32;
struct ReduceMulti {
  const float* gwp[kReduceMulti];
  float* gw[kReduceMulti];
  int cout[kReduceMulti], cin[kReduceMulti], taps[kReduceMulti], nslabs[kReduceMulti], tall[kReduceMulti];
};
__global__ __launch_bounds__(256) void reduce_unpack_wgrad_multi_kernel(ReduceMulti m) {
  const int j = blockIdx.y;
  const float* __restrict__ gwp = m.gwp[j];
  float* __restrict__ gw = m.gw[j];
  const int Cout = m.cout[j], Cin = m.cin[j], T = m.taps[j], nslabs = m.nslabs[j];
  const int CinP = (Cin + 3) / 4 * 4, CoutP = (Cout + 15) / 16 * 16;
  const long pe = (long)T * CinP * CoutP;
  __shared__ float part[16][17];
  if (m.tall[j]) {
    const int le = threadIdx.x & 15, ls = threadIdx.x >> 4;
    for (long base = (long)blockIdx.x * 16; base < pe; base += (long)gridDim.x * 16) {
      const long idx = base + le;
      float acc = 0.f;
      if (idx < pe) {
#pragma unroll 8
        for (int s = ls; s < nslabs; s += 16) acc += gwp[s * pe + idx];
      }
      __syncthreads();
      part[ls][le] = acc;
      __syncthreads();
      if (ls == 0 && idx < pe) {
#pragma unroll
        for (int k = 1; k < 16; ++k) acc += part[k][le];
        const int co = (int)(idx % CoutP);
        const long t = idx / CoutP;
        const int ci = (int)(t % CinP), tap = (int)(t / CinP);
        if (co < Cout && ci < Cin) gw[((long)co * Cin + ci) * T + tap] = acc;
      }
    }
    return;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float (*part4)[64] = reinterpret_cast<float (*)[64]>(&part[0][0]);      // [4][64] inside the 16 x 17 array
  for (long base = (long)blockIdx.x * 64; base < pe; base += (long)gridDim.x * 64) {
    const long idx = base + lane;
    float acc = 0.f;
    if (idx < pe) {
      // (eight loads in flight per lane, added in slab order: the sum is the chain's, the latency is not)
#pragma unroll 8
      for (int s = wid; s < nslabs; s += 4) acc += gwp[s * pe + idx];
    }
    __syncthreads();
    part4[wid][lane] = acc;
    __syncthreads();
    if (wid == 0 && idx < pe) {
      const float v = ((part4[0][lane] + part4[1][lane]) + part4[2][lane]) + part4[3][lane];
      const int co = (int)(idx % CoutP);
      const long t = idx / CoutP;
      const int ci = (int)(t % CinP), tap = (int)(t / CinP);
      if (co < Cout && ci < Cin) gw[((long)co * Cin + ci) * T + tap] = v;
    }
  }
}

// y[b][c][v] = bias[c] + sum_ks slab[ks][b][c][v]   (fixed order; per = Cout*DHW elements per batch item)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                            float* __restrict__ y, int ksplit, int n, long per, long DHW, long y_bstride) {
  const long total = (long)n * per;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / per, r = e - b * per;
    float acc = bias ? bias[r / DHW] : 0.f;
    for (int k = 0; k < ksplit; ++k) acc += slab[((long)k * n + b) * per + r];
    y[b * y_bstride + r] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Input halo staging.  One "pass" of the workgroup covers CPP channels x PZ x PY rows x Q items per row
// (Q = PXL/4 float4 when VEC, PXL scalars otherwise).  A thread owns up to JXMAX items of a pass; their
// (global offset, LDS offset, channel, in-bounds) are pass-invariant and computed once per tile.
// ---------------------------------------------------------------------------------------------------------
constexpr int kJX = 4;   // max items per thread per pass
constexpr int kNP = 4;   // max passes per chunk (KC / CPP)

template <int KS, int NT, bool VEC>
struct HaloStager {
  int goff[kJX];   // element offset inside the chunk's first channel group; -1 = out of the volume / no item
  int loff[kJX];
  int chan[kJX];

  __device__ __forceinline__ void setup(const ConvGeom& g, int z0, int y0, int x0) {
    constexpr int PAD = KS / 2;
    constexpr int E = VEC ? 4 : 1;
    const int PZ = g.TZ + KS - 1, PY = g.TY + KS - 1;
    const int Q = g.PXL / E;
    const int IP = g.CPP * PZ * PY * Q;
    const long HW = (long)g.H * g.W;
#pragma unroll
    for (int j = 0; j < kJX; ++j) {
      const int it = threadIdx.x + j * NT;
      goff[j] = -1; loff[j] = -1; chan[j] = 0;
      if (it < IP) {
        const int q = it % Q;
        int r = it / Q;
        const int py = r % PY;
        r /= PY;
        const int pz = r % PZ;
        const int c = r / PZ;
        const int gx = x0 - g.HX + q * E, gy = y0 - PAD + py, gz = z0 - PAD + pz;
        chan[j] = c;
        loff[j] = c * g.PS + (pz * PY + py) * g.PXL + q * E;
        const bool ok = gz >= 0 && gz < g.D && gy >= 0 && gy < g.H && gx >= 0 && gx + (E - 1) < g.W;
        if (ok) goff[j] = (int)((long)c * g.D * HW + gz * HW + (long)gy * g.W + gx);
      }
    }
  }
};

template <bool VEC> struct StageVal { float v[VEC ? 4 : 1]; };

template <int KS, int NT, bool VEC>
__device__ __forceinline__ void halo_load(const HaloStager<KS, NT, VEC>& st, StageVal<VEC> (&val)[kNP][kJX],
                                          const float* __restrict__ xb, int c0, const ConvGeom& g) {
  const long DHW = (long)g.D * g.H * g.W;
  const int np = g.KC / g.CPP;
#pragma unroll
  for (int p = 0; p < kNP; ++p) {
#pragma unroll
    for (int j = 0; j < kJX; ++j) {
      if (VEC) { val[p][j].v[0] = 0.f; val[p][j].v[1] = 0.f; val[p][j].v[2] = 0.f; val[p][j].v[3] = 0.f; }
      else val[p][j].v[0] = 0.f;
      if (p < np && st.goff[j] >= 0 && (c0 + p * g.CPP + st.chan[j]) < g.Cin) {
        const float* src = xb + (long)(c0 + p * g.CPP) * DHW + st.goff[j];
        if (VEC) {
          const float4 t = *reinterpret_cast<const float4*>(src);
          val[p][j].v[0] = t.x; val[p][j].v[1] = t.y; val[p][j].v[2] = t.z; val[p][j].v[3] = t.w;
        } else {
          val[p][j].v[0] = *src;
        }
      }
    }
  }
}

template <int KS, int NT, bool VEC>
__device__ __forceinline__ void halo_store(const HaloStager<KS, NT, VEC>& st, const StageVal<VEC> (&val)[kNP][kJX], float* Xs,
                                           const ConvGeom& g) {
  const int np = g.KC / g.CPP;
#pragma unroll
  for (int p = 0; p < kNP; ++p) {
#pragma unroll
    for (int j = 0; j < kJX; ++j) {
      if (p < np && st.loff[j] >= 0) {
        float* dst = Xs + p * g.CPP * g.PS + st.loff[j];
        if (VEC) *reinterpret_cast<float4*>(dst) = make_float4(val[p][j].v[0], val[p][j].v[1], val[p][j].v[2], val[p][j].v[3]);
        else *dst = val[p][j].v[0];
      }
    }
  }
}

// Packed-weight staging: rows (tap, c) of NB floats, 16 B per item.
constexpr int kWX = 7;  // max float4 items per thread per chunk

// column swizzle for 32-float rows: odd rows swap their 16-column halves, so the four k-rows an MFMA B-read touches
// (lanes 0-15 / 16-31 read rows k, k+1) fall on disjoint LDS banks without padding the row to 48 floats.
template <int NB>
__device__ __forceinline__ int wswz(int row, int col) { return NB == 32 ? (col ^ ((row & 1) << 4)) : col; }

template <int T, int NB, int NBP, int NT>
struct WeightStager {
  int goff[kWX];
  int loff[kWX];
  __device__ __forceinline__ void setup(const ConvGeom& g, int n0) {
    const int total = T * g.KC * (NB / 4);
#pragma unroll
    for (int i = 0; i < kWX; ++i) {
      const int it = threadIdx.x + i * NT;
      goff[i] = -1; loff[i] = -1;
      if (it < total) {
        const int n4 = it % (NB / 4), row = it / (NB / 4);
        const int tap = row / g.KC, c = row - tap * g.KC;
        loff[i] = row * NBP + wswz<NB>(row, n4 * 4);
        if (n0 + n4 * 4 < g.CoutP) goff[i] = (tap * g.CinP + c) * g.CoutP + n0 + n4 * 4;
      }
    }
  }
  __device__ __forceinline__ void load(float4 (&val)[kWX], const float* __restrict__ wp, int c0, const ConvGeom& g) const {
#pragma unroll
    for (int i = 0; i < kWX; ++i) {
      val[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (goff[i] >= 0) val[i] = *reinterpret_cast<const float4*>(wp + (long)c0 * g.CoutP + goff[i]);
    }
  }
  __device__ __forceinline__ void store(const float4 (&val)[kWX], float* Ws) const {
#pragma unroll
    for (int i = 0; i < kWX; ++i)
      if (loff[i] >= 0) *reinterpret_cast<float4*>(Ws + loff[i]) = val[i];
  }
};

template <int KS, int NBT, int MV, int WAVES, bool VEC>
__global__ __launch_bounds__(WAVES * 64) void conv3d_mfma_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                                     ConvGeom g) {
  constexpr int T = KS * KS * KS;
  constexpr int PAD = KS / 2;
  constexpr int NB = NBT * 16;
  constexpr int NBP = NB;  // row pitch 16 / 48 is == 16 (mod 32); the 32-wide rows are XOR-swizzled (wswz) instead of padded
  constexpr int NT = WAVES * 64;
  ICL_DYN_LDS(float, lds);
  float* Xs = lds;
  float* Ws = lds + g.KC * g.PS;
  const int PY = g.TY + KS - 1;
  const int ntiles = g.ntx * g.nty * g.ntz;
  const int bt = blockIdx.x % ntiles;
  const int ks = blockIdx.x / ntiles;
  const int x0 = (bt % g.ntx) * g.TX;
  const int y0 = ((bt / g.ntx) % g.nty) * g.TY;
  const int z0 = (bt / (g.ntx * g.nty)) * g.TZ;
  const int n0 = blockIdx.y * NB;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const float* xb = x + (long)blockIdx.z * g.x_bstride;
  float* yb = g.ksplit > 1 ? g.slab + ((long)ks * gridDim.z + blockIdx.z) * g.Cout * DHW : y + (long)blockIdx.z * g.y_bstride;
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
    vbase[m] = (tz * PY + ty) * g.PXL + tx + (g.HX - PAD) + lq * g.PS;
  }
  f32x4 acc[MV][NBT];
#pragma unroll
  for (int m = 0; m < MV; ++m)
#pragma unroll
    for (int j = 0; j < NBT; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  HaloStager<KS, NT, VEC> hs;
  hs.setup(g, z0, y0, x0);
  WeightStager<T, NB, NBP, NT> wst;
  wst.setup(g, n0);
  StageVal<VEC> xv[kNP][kJX];
  float4 wv[kWX];

  const int nchunks = g.CinP / g.KC;
  int ci = ks;
  if (ci < nchunks) {
    halo_load<KS, NT, VEC>(hs, xv, xb, ci * g.KC, g);
    wst.load(wv, wp, ci * g.KC, g);
  }
  for (; ci < nchunks; ci += g.ksplit) {
    __syncthreads();  // every wave is done reading the previous chunk
    halo_store<KS, NT, VEC>(hs, xv, Xs, g);
    wst.store(wv, Ws);
    __syncthreads();
    if (ci + g.ksplit < nchunks) {  // prefetch the next chunk into registers; it lands while the MFMAs run
      halo_load<KS, NT, VEC>(hs, xv, xb, (ci + g.ksplit) * g.KC, g);
      wst.load(wv, wp, (ci + g.ksplit) * g.KC, g);
    }
    for (int tap = 0; tap < T; ++tap) {
      const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
      const int tapoff = (dz * PY + dy) * g.PXL + dx;
      for (int cc = 0; cc < g.KC; cc += 4) {
        const int krow = tap * g.KC + cc + lq;
        float b[NBT];
#pragma unroll
        for (int j = 0; j < NBT; ++j) b[j] = Ws[krow * NBP + wswz<NB>(krow, j * 16 + lr)];
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
    const float bv = (bias && g.ksplit == 1) ? bias[co] : 0.f;
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
          if (gz < g.D && gy < g.H && gx < g.W) {
            yc[gz * HW + (long)gy * g.W + gx] = acc[m][j][r] + bv;
          }
        }
      }
    }
  }
}

// dW partials: grid.x = spatial split, grid.y = (CoutP/16) * ceil(CinP/16), grid.z = batch.
// Requires TX % 4 == 0 and KC == 16.  Gs pitch MTP and plane stride PS are == 2 (mod 32) in the scalar layout;
// with VEC staging PS is a multiple of 4 with PS/2 odd... see host: PS == 4*odd keeps 2-way conflicts at most.
template <int KS, int NTW, int WAVES, bool VEC>
__global__ __launch_bounds__(WAVES * 64) void conv3d_mfma_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                       float* __restrict__ gwp, ConvGeom g, int MTP) {
  constexpr int T = KS * KS * KS;
  constexpr int PAD = KS / 2;
  constexpr int NT = WAVES * 64;
  ICL_DYN_LDS(float, lds);
  float* Xs = lds;                // [16][PS]
  float* Gs = lds + 16 * g.PS;    // [16][MTP]
  const int PY = g.TY + KS - 1;
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
    const int tap = (T >= WAVES) ? wid + t * WAVES : 0;
    const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
    tapoff[t] = (dz * PY + dy) * g.PXL + dx + (g.HX - PAD);
  }
  f32x4 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  HaloStager<KS, NT, VEC> hs;
  StageVal<VEC> xv[kNP][kJX];
  // dY tile items: 16 channels x MT/E elements; up to 4 per thread (MT <= 256 when VEC, checked on the host)
  constexpr int E = VEC ? 4 : 1;
  constexpr int kGX = VEC ? 4 : 16;

  for (int bt = blockIdx.x; bt < ntiles; bt += gridDim.x) {
    const int x0 = (bt % g.ntx) * g.TX;
    const int y0 = ((bt / g.ntx) % g.nty) * g.TY;
    const int z0 = (bt / (g.ntx * g.nty)) * g.TZ;
    hs.setup(g, z0, y0, x0);
    halo_load<KS, NT, VEC>(hs, xv, xb, c0, g);
    StageVal<VEC> gv[kGX];
    int gl[kGX];
#pragma unroll
    for (int i = 0; i < kGX; ++i) {
      const int it = threadIdx.x + i * NT;
      gl[i] = -1;
      if (VEC) { gv[i].v[0] = 0.f; gv[i].v[1] = 0.f; gv[i].v[2] = 0.f; gv[i].v[3] = 0.f; }
      else gv[i].v[0] = 0.f;
      if (it < 16 * (MT / E)) {
        const int vt = (it % (MT / E)) * E, co = it / (MT / E);
        const int tx = vt % g.TX, t2 = vt / g.TX;
        const int ty = t2 % g.TY, tz = t2 / g.TY;
        const int gz = z0 + tz, gyy = y0 + ty, gx = x0 + tx;
        gl[i] = co * MTP + vt;
        if (co0 + co < g.Cout && gz < g.D && gyy < g.H && gx + (E - 1) < g.W) {
          const float* src = gb + (long)(co0 + co) * DHW + gz * HW + (long)gyy * g.W + gx;
          if (VEC) {
            const float4 t4 = *reinterpret_cast<const float4*>(src);
            gv[i].v[0] = t4.x; gv[i].v[1] = t4.y; gv[i].v[2] = t4.z; gv[i].v[3] = t4.w;
          } else {
            gv[i].v[0] = *src;
          }
        }
      }
    }
    __syncthreads();  // previous tile fully consumed
    halo_store<KS, NT, VEC>(hs, xv, Xs, g);
#pragma unroll
    for (int i = 0; i < kGX; ++i) {
      if (gl[i] >= 0) {
        if (VEC) {  // MTP is even, not a multiple of 4: two 8-byte stores
          *reinterpret_cast<float2*>(Gs + gl[i]) = make_float2(gv[i].v[0], gv[i].v[1]);
          *reinterpret_cast<float2*>(Gs + gl[i] + 2) = make_float2(gv[i].v[2], gv[i].v[3]);
        } else {
          Gs[gl[i]] = gv[i].v[0];
        }
      }
    }
    __syncthreads();
    if (T >= WAVES) {
      for (int k0 = 0; k0 < MT; k0 += 4) {
        const int tx = k0 % g.TX, t2 = k0 / g.TX;
        const int ty = t2 % g.TY, tz = t2 / g.TY;
        const int vb = (tz * PY + ty) * g.PXL + tx + lq + lr * g.PS;
        const float a = Gs[lr * MTP + k0 + lq];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (wid + t * WAVES < T) {
            const float b = Xs[vb + tapoff[t]];
            acc[t] = icl_mfma_16x16x4(a, b, acc[t]);
          }
        }
      }
    } else {  // fewer taps than waves (1x1x1): the waves split the voxels instead
      for (int k0 = wid * 4; k0 < MT; k0 += 4 * WAVES) {
        const int tx = k0 % g.TX, t2 = k0 / g.TX;
        const int ty = t2 % g.TY, tz = t2 / g.TY;
        const int vb = (tz * PY + ty) * g.PXL + tx + lq + lr * g.PS;
        const float a = Gs[lr * MTP + k0 + lq];
        const float b = Xs[vb + tapoff[0]];
        acc[0] = icl_mfma_16x16x4(a, b, acc[0]);
      }
    }
  }
  // D[row = co0 + lq*4 + r][col = c0 + lr]
  const int ci = c0 + lr;
  if (ci < g.CinP) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int tap = (T >= WAVES) ? wid + t * WAVES : 0;
      if (tap >= T) continue;
      // every (split, batch) workgroup owns one slab of packed partial sums: plain 16-byte stores, no atomics;
      // reduce_unpack_wgrad_kernel adds the slabs in a fixed order (bitwise reproducible gradients)
      // (with fewer taps than waves the waves hold partial sums of the SAME tap: one slab per wave)
      const long slab = ((long)blockIdx.z * gridDim.x + blockIdx.x) * (T >= WAVES ? 1 : WAVES) + (T >= WAVES ? 0 : wid);
      float* dst = gwp + slab * ((long)T * g.CinP * g.CoutP) + ((long)tap * g.CinP + ci) * g.CoutP + co0 + lq * 4;
      *reinterpret_cast<float4*>(dst) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
  }
}

}  // namespace icl
