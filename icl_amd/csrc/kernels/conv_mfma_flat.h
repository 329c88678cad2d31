// conv_mfma_flat.h — compile-time-tile 3x3x3 forward/dgrad kernel for NARROW volumes (W = 24, 12, 6).
//
// conv_mfma_static.h fixes TX = 16 so that one MFMA row group is one x-row; on the deep U-Net levels that wastes
// 25-62 % of every MFMA on masked voxels.  Here the tile spans the full row (TX = W) and the 16 voxels of a row group
// are 16 consecutive voxels of the FLATTENED tile, so a group may wrap over rows: each group keeps its own LDS base
// register (MV registers per wave) while every (tap, channel) displacement is still an instruction immediate.
// Same layout, staging, numerics and Cin split as the other conv kernels.  VEC = 16-byte staging (W % 4 == 0).
#pragma once

namespace icl {

template <int TZ_, int TY_, int TX_, int KC_, bool VEC_>
struct FlatTile {
  static constexpr int T = 27;
  static constexpr bool VEC = VEC_;
  static constexpr int TZ = TZ_, TY = TY_, TX = TX_, KC = KC_;
  static constexpr int E = VEC ? 4 : 1;
  static constexpr int HX = VEC ? 4 : 1, PXL = TX + 2 * HX;
  static constexpr int PZ = TZ + 2, PY = TY + 2;
  static constexpr int Q = PXL / E;
  static constexpr int PER_CH = PZ * PY * Q;
  static constexpr int PS = pad_to_mod(PZ * PY * PXL, 16, 32);
  static constexpr int MT = TZ * TY * TX;
  static constexpr int G = (MT + 15) / 16;
  static constexpr int WAVES = 4, NT = 256;
  static constexpr int MV = (G + WAVES - 1) / WAVES;
  static constexpr int CPP = 2;
  static constexpr int NP = KC / CPP;
  static constexpr int JX = (CPP * PER_CH + NT - 1) / NT;
  static_assert(KC % 4 == 0 && KC % CPP == 0, "bad channel chunk");
  static_assert(!VEC || (TX % 4 == 0), "vector staging needs TX % 4 == 0");
};

template <bool VEC> struct StageReg;
template <> struct StageReg<true> {
  float4 v;
  __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct StageReg<false> {
  float v;
  __device__ __forceinline__ void zero() { v = 0.f; }
  __device__ __forceinline__ void load(const float* p) { v = *p; }
  __device__ __forceinline__ void store(float* p) const { *p = v; }
};

template <int NBT, class TC>
__global__ __launch_bounds__(256) void conv3d_mfma_fwd_flat_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                   const float* __restrict__ bias, float* __restrict__ y,
                                                                   ConvGeom g) {
  constexpr int T = TC::T, KC = TC::KC, PS = TC::PS, PXL = TC::PXL, PY = TC::PY, PZ = TC::PZ, MV = TC::MV, MT = TC::MT;
  constexpr int NB = NBT * 16, NBP = NB, NT = TC::NT, E = TC::E;
  constexpr int NP = TC::NP, JX = TC::JX, CPP = TC::CPP, Q = TC::Q;
  constexpr int WITEMS = T * KC * (NB / 4);
  constexpr int WX = (WITEMS + NT - 1) / NT;
  ICL_DYN_LDS(float, lds);
  float* Xs = lds;
  float* Ws = lds + KC * PS;
  const int ntiles = g.ntx * g.nty * g.ntz;
  const int bt = blockIdx.x % ntiles;
  const int ks = blockIdx.x / ntiles;
  const int x0 = (bt % g.ntx) * TC::TX;
  const int y0 = ((bt / g.ntx) % g.nty) * TC::TY;
  const int z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
  const int n0 = blockIdx.y * NB;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const float* xb = x + (long)blockIdx.z * g.x_bstride;
  float* yb = g.ksplit > 1 ? g.slab + ((long)ks * gridDim.z + blockIdx.z) * g.Cout * DHW : y + (long)blockIdx.z * g.y_bstride;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lq = lane >> 4, lr = lane & 15;

  int xg[JX], xl[JX], xc[JX];
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int it = threadIdx.x + j * NT;
    xg[j] = -1; xl[j] = -1; xc[j] = 0;
    if (it < CPP * TC::PER_CH) {
      const int q = it % Q;
      int r = it / Q;
      const int py = r % PY;
      r /= PY;
      const int pz = r % PZ;
      const int c = r / PZ;
      const int gx = x0 - TC::HX + q * E, gy = y0 - 1 + py, gz = z0 - 1 + pz;
      xc[j] = c;
      xl[j] = c * PS + (pz * PY + py) * PXL + q * E;
      if (gz >= 0 && gz < g.D && gy >= 0 && gy < g.H && gx >= 0 && gx + (E - 1) < g.W)
        xg[j] = (int)((long)c * DHW + gz * HW + (long)gy * g.W + gx);
    }
  }
  int wg[WX], wl[WX];
#pragma unroll
  for (int i = 0; i < WX; ++i) {
    const int it = threadIdx.x + i * NT;
    wg[i] = -1; wl[i] = -1;
    if (it < WITEMS) {
      const int n4 = it % (NB / 4), row = it / (NB / 4);
      const int tap = row / KC, c = row - tap * KC;
      wl[i] = row * NBP + wswz<NB>(row, n4 * 4);
      if (n0 + n4 * 4 < g.CoutP) wg[i] = (tap * g.CinP + c) * g.CoutP + n0 + n4 * 4;
    }
  }
  StageReg<TC::VEC> xv[NP][JX];
  float4 wv[WX];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < JX; ++j) {
        xv[p][j].zero();
        if (xg[j] >= 0 && (c0 + p * CPP + xc[j]) < g.Cin) xv[p][j].load(xb + (long)(c0 + p * CPP) * DHW + xg[j]);
      }
#pragma unroll
    for (int i = 0; i < WX; ++i) {
      wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wg[i] >= 0) wv[i] = *reinterpret_cast<const float4*>(wp + (long)c0 * g.CoutP + wg[i]);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < JX; ++j)
        if (xl[j] >= 0) xv[p][j].store(Xs + p * CPP * PS + xl[j]);
#pragma unroll
    for (int i = 0; i < WX; ++i)
      if (wl[i] >= 0) *reinterpret_cast<float4*>(Ws + wl[i]) = wv[i];
  };

  // one LDS base per row group (voxel (tz,ty,tx) of the flattened tile), tap/channel displacements are immediates
  int vbase[MV];
#pragma unroll
  for (int m = 0; m < MV; ++m) {
    int vt = (wid * MV + m) * 16 + lr;
    if (vt >= MT) vt = MT - 1;
    const int tx = vt % TC::TX, t2 = vt / TC::TX;
    const int ty = t2 % TC::TY, tz = t2 / TC::TY;
    vbase[m] = lq * PS + (tz * PY + ty) * PXL + tx + (TC::HX - 1);
  }
  int bbase[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) bbase[j] = lq * NBP + wswz<NB>(lq, j * 16 + lr);

  f32x4 acc[MV][NBT];
#pragma unroll
  for (int m = 0; m < MV; ++m)
#pragma unroll
    for (int j = 0; j < NBT; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = g.CinP / KC;
  int ci = ks;
  if (ci < nchunks) load_chunk(ci * KC);
  for (; ci < nchunks; ci += g.ksplit) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    if (ci + g.ksplit < nchunks) load_chunk((ci + g.ksplit) * KC);
    // software pipeline over the T * KC/4 k-steps of the chunk: the operands of step i+1 are read from LDS before the MFMAs of
    // step i are issued (see conv3d_mfma_wgrad_static_kernel)
    constexpr int STEPS = T * (KC / 4);
    auto load_ab = [&](int step, float (&av)[MV], float (&bv)[NBT]) {
      const int tap = step / (KC / 4), cc = (step % (KC / 4)) * 4;
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
#pragma unroll
      for (int j = 0; j < NBT; ++j) bv[j] = Ws[bbase[j] + (tap * KC + cc) * NBP];
#pragma unroll
      for (int m = 0; m < MV; ++m) av[m] = Xs[vbase[m] + cc * PS + (dz * PY + dy) * PXL + dx];
    };
    float a[MV], an[MV], b[NBT], bn[NBT];
    load_ab(0, a, b);
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
      if (step + 1 < STEPS) load_ab(step + 1, an, bn);
      ICL_SCHED_BARRIER();
#pragma unroll
      for (int m = 0; m < MV; ++m)
#pragma unroll
        for (int j = 0; j < NBT; ++j) acc[m][j] = icl_mfma_16x16x4(a[m], b[j], acc[m][j]);
      ICL_SCHED_BARRIER();
#pragma unroll
      for (int m = 0; m < MV; ++m) a[m] = an[m];
#pragma unroll
      for (int j = 0; j < NBT; ++j) b[j] = bn[j];
    }
  }

  const bool vec = TC::VEC;
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
        const int tx = vt0 % TC::TX, t2 = vt0 / TC::TX;
        const int gz = z0 + t2 / TC::TY, gy = y0 + t2 % TC::TY, gx = x0 + tx;
        if (gz < g.D && gy < g.H && gx < g.W)
          *reinterpret_cast<float4*>(yc + gz * HW + (long)gy * g.W + gx) =
              make_float4(acc[m][j][0] + bv, acc[m][j][1] + bv, acc[m][j][2] + bv, acc[m][j][3] + bv);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int vt = vt0 + r;
          if (vt >= MT) continue;
          const int tx = vt % TC::TX, t2 = vt / TC::TX;
          const int gz = z0 + t2 / TC::TY, gy = y0 + t2 % TC::TY, gx = x0 + tx;
          if (gz < g.D && gy < g.H && gx < g.W) {
            yc[gz * HW + (long)gy * g.W + gx] = acc[m][j][r] + bv;
          }
        }
      }
    }
  }
}

}  // namespace icl
