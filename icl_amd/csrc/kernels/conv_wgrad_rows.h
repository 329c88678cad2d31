// conv_wgrad_rows.h — weight gradient of the 3x3x3 convolution, "row window" form (autograd of networks/utils.py:104,107).
//
// dW[tap][ci][co] = sum_v gy[co][v] * x[ci][v + tap] on v_mfma_f32_16x16x4_f32: M = 16 output channels, N = 16 input channels,
// K = voxels.  The predecessor (conv_mfma_static.h) read one LDS dword per MFMA and operand and was bound by LDS-read LATENCY
// (PMC: 1.19 LDS instructions per MFMA, waves parked 56 % of their life).  Here the MFMA k label is chosen so that wide reads
// feed many MFMAs: in MFMA s of a 16-voxel row, lane group lq supplies voxel 4 lq + s.  Then
//   A (gy):  the four MFMAs of a row take the four components of ONE ds_read_b128 (voxels 4 lq .. 4 lq + 3 of channel lr);
//   B (x):   for a fixed (dz, dy) the three dx taps x four MFMAs read x positions 4 lq - 1 .. 4 lq + 4 of the halo row: six
//            consecutive floats = one ds_read_b128 + two ds_read_b64, kept in registers for 12 MFMAs per cout block.
// => 0.28 LDS instructions per MFMA with one cout block, 0.1 with three.  A wave owns one dz plane of taps (9 taps: three
// (dy) triples) and a share of the tile's rows; the RG waves of a dz group add their accumulators through LDS at the end of the
// workgroup's tile run (fixed order), so a workgroup still writes ONE packed partial-sum slab (reduce_unpack_wgrad_kernel).
#pragma once

namespace icl {

template <int TZ_, int TY_>
struct WgradRowsTile {
  static constexpr int TZ = TZ_, TY = TY_, TX = 16, HX = 4, PXL = TX + 2 * HX, PZ = TZ + 2, PY = TY + 2, Q = PXL / 4;
  static constexpr int PER_CH = PZ * PY * Q;                       // float4 items per channel of the halo tile
  static constexpr int PS = pad_to_mod(PZ * PY * PXL, 4, 64);      // channel plane pitch == 4 (mod 64): 16-byte aligned rows, the 16
  static constexpr int MT = TZ * TY * TX;                          // planes of a b128 fragment read land in (almost) distinct slots
  static constexpr int MTP = pad_to_mod(MT, 4, 64);
  static constexpr int ROWS = TZ * TY;
};

// grid (nsplit, cout-group x cin-block pairs, batch), block 192 * RG threads: wave = (dz group tgp = wave % 3, row group wave / 3).
// dynamic LDS: (16 PS + 16 NCB MTP) floats, reused for the final cross-wave sum (needs 3 RG * RG KiB).
template <class TC, int NCB, int RG>
__global__ __launch_bounds__(192 * RG) void conv3d_wgrad_rows_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                     float* __restrict__ gwp, ConvGeom g) {
  constexpr int T = 27, TGN = 3, NTAP = 9, WV = TGN * RG, NT = 64 * WV, MB = 16 * NCB;
  constexpr int PS = TC::PS, PXL = TC::PXL, PY = TC::PY, PZ = TC::PZ, MT = TC::MT, MTP = TC::MTP, Q = TC::Q;
  constexpr int RPW = TC::ROWS / RG;                                // rows per wave and tile
  static_assert(TC::ROWS % RG == 0 && (TC::TY % RPW == 0 || RPW % TC::TY == 0), "a wave's rows must tile the (z, y) grid");
  constexpr int XI = (16 * TC::PER_CH + NT - 1) / NT;              // x float4 items per thread
  constexpr int GI = (MB * (MT / 4) + NT - 1) / NT;                // gy float4 items per thread
  ICL_DYN_LDS(float, lds);
  float* Xs = lds;
  float* Gs = lds + 16 * PS;
  const int ncin = (g.CinP + 15) / 16;
  const int co0 = (blockIdx.y / ncin) * MB;
  const int c0 = (blockIdx.y % ncin) * 16;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const float* xb = x + (long)blockIdx.z * g.x_bstride;
  const float* gb = gy + (long)blockIdx.z * g.y_bstride;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int tgp = wid % TGN, rg = wid / TGN;
  const int lq = lane >> 4, lr = lane & 15;
  const int ntiles = g.ntz * g.nty * g.ntx;

  // tile-invariant staging tables
  int xpz[XI], xpy[XI], xq[XI], xch[XI], xl[XI];
#pragma unroll
  for (int j = 0; j < XI; ++j) {
    const int it = threadIdx.x + j * NT;
    xl[j] = -1; xpz[j] = 0; xpy[j] = 0; xq[j] = 0; xch[j] = 0;
    if (it < 16 * TC::PER_CH) {
      const int q = it % Q;
      int r = it / Q;
      const int py = r % PY;
      r /= PY;
      xpz[j] = r % PZ; xpy[j] = py; xq[j] = q; xch[j] = r / PZ;
      xl[j] = xch[j] * PS + (xpz[j] * PY + py) * PXL + q * 4;
    }
  }
  int gco[GI], gvt[GI];
#pragma unroll
  for (int i = 0; i < GI; ++i) {
    const int it = threadIdx.x + i * NT;
    gvt[i] = (it % (MT / 4)) * 4;
    gco[i] = it < MB * (MT / 4) ? it / (MT / 4) : -1;
  }
  float4 xv[XI], gv[GI];
  auto load_tile = [&](int bt) {
    const int x0 = (bt % g.ntx) * TC::TX;
    const int y0 = ((bt / g.ntx) % g.nty) * TC::TY;
    const int z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int gx = x0 - TC::HX + xq[j] * 4, gyy = y0 - 1 + xpy[j], gz = z0 - 1 + xpz[j];
      const int ch = c0 + xch[j];
      const bool ok = xl[j] >= 0 && ch < g.Cin && gz >= 0 && gz < g.D && gyy >= 0 && gyy < g.H && gx >= 0 && gx + 3 < g.W;
      xv[j] = *reinterpret_cast<const float4*>(xb + (ok ? (long)ch * DHW + gz * HW + (long)gyy * g.W + gx : 0L));
      if (!ok) xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      const int vt = gvt[i];
      const int tx = vt % TC::TX, t2 = vt / TC::TX;
      const int gz = z0 + t2 / TC::TY, gyy = y0 + t2 % TC::TY, gx = x0 + tx;
      const bool ok = gco[i] >= 0 && co0 + gco[i] < g.Cout && gz < g.D && gyy < g.H && gx + 3 < g.W;
      gv[i] = *reinterpret_cast<const float4*>(gb + (ok ? (long)(co0 + gco[i]) * DHW + gz * HW + (long)gyy * g.W + gx : 0L));
      if (!ok) gv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < XI; ++j)
      if (xl[j] >= 0) *reinterpret_cast<float4*>(Xs + xl[j]) = xv[j];
#pragma unroll
    for (int i = 0; i < GI; ++i)
      if (gco[i] >= 0) *reinterpret_cast<float4*>(Gs + gco[i] * MTP + gvt[i]) = gv[i];
  };

  // operand bases of this wave: first row (z0w, y0w) of its RPW rows; everything else is a compile-time displacement
  const int r0 = rg * RPW;
  const int z0w = r0 / TC::TY, y0w = r0 % TC::TY;
  const float* xa = Xs + lr * PS + ((z0w + tgp) * PY + y0w) * PXL + (TC::HX - 1) + 4 * lq;   // (dz = tgp, dy = 0, x = 4 lq - 1)
  const float* ga = Gs + lr * MTP + 16 * r0 + 4 * lq;
  f32x4 acc[NCB][NTAP];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lbx = g.remap ? xcd_chunked(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int bt = (int)((long)lbx * ntiles / gridDim.x);
  const int bt_end = (int)((long)(lbx + 1) * ntiles / gridDim.x);
  if (bt < bt_end) load_tile(bt);
  for (; bt < bt_end; ++bt) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (bt + 1 < bt_end) load_tile(bt + 1);
    // one row per iteration (rolled for more than one cout block: unrolled, the scheduler hoists the operand reads of all rows
    // and spills)
#pragma unroll(NCB == 1 ? RPW : 1)
    for (int rr = 0; rr < RPW; ++rr) {
      const int roff = (RPW <= TC::TY) ? rr * PXL : ((rr / TC::TY) * PY * PXL + (rr % TC::TY) * PXL);
      float4 a[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) a[cb] = *reinterpret_cast<const float4*>(ga + cb * 16 * MTP + 16 * rr);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const float* p = xa + roff + dy * PXL;
        const float2 lo = *reinterpret_cast<const float2*>(p - 1);      // (., x[4 lq - 1])
        const float4 mid = *reinterpret_cast<const float4*>(p + 1);     // x[4 lq .. 4 lq + 3]
        const float2 hi = *reinterpret_cast<const float2*>(p + 5);      // (x[4 lq + 4], .)
        const float w[6] = {lo.y, mid.x, mid.y, mid.z, mid.w, hi.x};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb][dy * 3 + dx] = icl_mfma_16x16x4(f4c(a[cb], s), w[s + dx], acc[cb][dy * 3 + dx]);
      }
    }
  }

  // ---- cross-wave sum of the RG row groups of each dz group, RG items per pass, then one packed slab per workgroup
  constexpr int NI = NCB * NTAP;
  float4* red = reinterpret_cast<float4*>(lds);
  const int ci = c0 + lr;
  float* slab = gwp + ((long)blockIdx.z * gridDim.x + blockIdx.x) * ((long)T * g.CinP * g.CoutP);
#pragma unroll
  for (int p0 = 0; p0 < NI; p0 += RG) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RG; ++k) {
      if (p0 + k < NI) {
        const f32x4 v = acc[(p0 + k) / NTAP][(p0 + k) % NTAP];
        red[(wid * RG + k) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();
    const int item = p0 + rg;                                       // the item this wave sums
    if (item < NI && ci < g.CinP) {
      float4 s = red[((0 * TGN + tgp) * RG + rg) * 64 + lane];
#pragma unroll
      for (int o = 1; o < RG; ++o) {
        const float4 v = red[((o * TGN + tgp) * RG + rg) * 64 + lane];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int cb = item / NTAP, tap = tgp * NTAP + item % NTAP;
      if (co0 + cb * 16 < g.CoutP) *reinterpret_cast<float4*>(slab + ((long)tap * g.CinP + ci) * g.CoutP + co0 + cb * 16 + lq * 4) = s;
    }
  }
}

}  // namespace icl
