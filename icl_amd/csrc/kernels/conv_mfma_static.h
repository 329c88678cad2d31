// conv_mfma_static.h — the 3x3x3 convolution kernels of conv_mfma.h specialised for a compile-time tile.
//
// Why: in the generic kernels every LDS operand address is "register + runtime stride", which costs one VALU add per
// ds_read and keeps the k-loop rolled; measured on MI355X (profiles/r1_pmc_conv.md) that loop issues ~5 VALU
// instructions per MFMA and the matrix pipe sits at 47 %.  With TZ/TY/TX/KC fixed at compile time the whole
// Cin-chunk (27 taps x KC/4 k-steps) unrolls into straight-line code whose ds_read offsets are instruction
// immediates: one base register per operand, no address arithmetic, and the scheduler is free to hoist the LDS
// reads of the next k-step above the MFMAs of the current one.
//
// Same data layout, GEMM view, staging scheme and numerics (v_mfma_f32_16x16x4_f32, k-ordered fmaf chain) as
// conv_mfma.h; TX is fixed to 16 so one MFMA row-group is one x-row of the tile and a wave's groups are whole rows.
// Requires W % 4 == 0 (16-byte staging).  Partial tiles at the volume border are masked.
#pragma once

namespace icl {

constexpr int pad_to_mod(int v, int mod, int of) { return (v % of == mod) ? v : pad_to_mod(v + 1, mod, of); }

template <int TZ_, int TY_, int KC_>
struct FwdTile {
  static constexpr int KS = 3, PAD = 1, T = 27;
  static constexpr int TZ = TZ_, TY = TY_, TX = 16, KC = KC_;
  static constexpr int HX = 4, PXL = TX + 2 * HX;
  static constexpr int PZ = TZ + 2, PY = TY + 2;
  static constexpr int Q = PXL / 4;
  static constexpr int PER_CH = PZ * PY * Q;            // float4 items per channel
  static constexpr int PS = pad_to_mod(PZ * PY * PXL, 16, 32);
  static constexpr int G = TZ * TY;                      // 16-voxel row groups per tile
  static constexpr int WAVES = 4, NT = 256;
  static constexpr int MV = G / WAVES;
  static constexpr int CPP = (KC >= 8) ? 2 : (PER_CH * KC <= 3 * NT ? KC : 2);  // channels per staging pass
  static constexpr int NP = KC / CPP;
  static constexpr int JX = (CPP * PER_CH + NT - 1) / NT;
  static_assert(G % WAVES == 0, "tile rows must split evenly over the waves");
  static_assert(MV % TY == 0 || TY % MV == 0, "a wave's rows must be whole z-slices or stay inside one");
  static_assert(KC % 4 == 0 && KC % CPP == 0, "bad channel chunk");
  // LDS offset of row group m of a wave relative to the wave's first row
  static constexpr int row_off(int m) { return (MV % TY == 0) ? ((m / TY) * PY * PXL + (m % TY) * PXL) : m * PXL; }
};

template <int NBT, class TC>
__global__ __launch_bounds__(256) void conv3d_mfma_fwd_static_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                                     ConvGeom g) {
  constexpr int T = TC::T, KC = TC::KC, PS = TC::PS, PXL = TC::PXL, PY = TC::PY, PZ = TC::PZ, MV = TC::MV;
  constexpr int NB = NBT * 16, NBP = NB, NT = TC::NT;
  constexpr int NP = TC::NP, JX = TC::JX, CPP = TC::CPP, Q = TC::Q;
  constexpr int WITEMS = T * KC * (NB / 4);
  constexpr int WX = (WITEMS + NT - 1) / NT;
  ICL_DYN_LDS(float, lds);
  float* Xs = lds;
  float* Ws = lds + KC * PS;
  const int ntiles = g.ntx * g.nty * g.ntz;
  const int bx = g.remap ? xcd_chunked(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int bt = bx % ntiles;
  const int ks = bx / ntiles;
  const int x0 = (bt % g.ntx) * TC::TX;
  const int y0 = ((bt / g.ntx) % g.nty) * TC::TY;
  const int z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
  const int n0 = blockIdx.y * NB;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const float* xb = x + (long)blockIdx.z * g.x_bstride;
  float* yb = g.ksplit > 1 ? g.slab + ((long)ks * gridDim.z + blockIdx.z) * g.Cout * DHW : y + (long)blockIdx.z * g.y_bstride;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lq = lane >> 4, lr = lane & 15;

  // ---- staging tables (tile-invariant, chunk-invariant) ----
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
      const int gx = x0 - TC::HX + q * 4, gy = y0 - 1 + py, gz = z0 - 1 + pz;
      xc[j] = c;
      xl[j] = c * PS + (pz * PY + py) * PXL + q * 4;
      if (gz >= 0 && gz < g.D && gy >= 0 && gy < g.H && gx >= 0 && gx + 3 < g.W)
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
  float4 xv[NP][JX];
  float4 wv[WX];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < JX; ++j) {
        xv[p][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xg[j] >= 0 && (c0 + p * CPP + xc[j]) < g.Cin)
          xv[p][j] = *reinterpret_cast<const float4*>(xb + (long)(c0 + p * CPP) * DHW + xg[j]);
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
        if (xl[j] >= 0) *reinterpret_cast<float4*>(Xs + p * CPP * PS + xl[j]) = xv[p][j];
#pragma unroll
    for (int i = 0; i < WX; ++i)
      if (wl[i] >= 0) *reinterpret_cast<float4*>(Ws + wl[i]) = wv[i];
  };

  // ---- operand base addresses: everything else is a compile-time offset ----
  const int wrow0 = wid * MV;  // first row group of this wave
  const int abase = lq * PS + (wrow0 / TC::TY) * PY * PXL + (wrow0 % TC::TY) * PXL + lr + (TC::HX - 1);
  int bbase[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) bbase[j] = lq * NBP + wswz<NB>(lq, j * 16 + lr);  // KC is even: row parity == lq parity

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
    const float* Xa = Xs + abase;
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      constexpr int dummy = 0;
      (void)dummy;
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
#pragma unroll
      for (int cc = 0; cc < KC; cc += 4) {
        float b[NBT];
#pragma unroll
        for (int j = 0; j < NBT; ++j) b[j] = Ws[bbase[j] + (tap * KC + cc) * NBP];
#pragma unroll
        for (int m = 0; m < MV; ++m) {
          const float a = Xa[cc * PS + TC::row_off(m) + (dz * PY + dy) * PXL + dx];
#pragma unroll
          for (int j = 0; j < NBT; ++j) acc[m][j] = icl_mfma_16x16x4(a, b[j], acc[m][j]);
        }
      }
    }
  }

  // ---- epilogue: lane holds rows x = lq*4 + r of row group (wid*MV + m), column co = n0 + j*16 + lr ----
  const int gx = x0 + lq * 4;
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int co = n0 + j * 16 + lr;
    if (co >= g.Cout) continue;
    const float bv = (bias && g.ksplit == 1) ? bias[co] : 0.f;
    float* yc = yb + (long)co * DHW;
#pragma unroll
    for (int m = 0; m < MV; ++m) {
      const int grp = wrow0 + m;
      const int gz = z0 + grp / TC::TY, gy = y0 + grp % TC::TY;
      if (gz < g.D && gy < g.H && gx < g.W) {
        float* dst = yc + gz * HW + (long)gy * g.W + gx;
        *reinterpret_cast<float4*>(dst) = make_float4(acc[m][j][0] + bv, acc[m][j][1] + bv, acc[m][j][2] + bv, acc[m][j][3] + bv);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// wgrad, tile 2 x 8 x 16 voxels, 16 output x 16 input channels x 27 taps per workgroup (taps round-robin over waves).
// ---------------------------------------------------------------------------------------------------------
template <int TZ_, int TY_>
struct WgradTileT {
  static constexpr int TZ = TZ_, TY = TY_, TX = 16, KC = 16;
  static constexpr int HX = 4, PXL = TX + 2 * HX, PZ = TZ + 2, PY = TY + 2, Q = PXL / 4;
  static constexpr int PER_CH = PZ * PY * Q;
  static constexpr int PS = pad_to_mod(PZ * PY * PXL, 2, 32);   // B reads: lanes lr -> planes, conflict-free
  static constexpr int MT = TZ * TY * TX, MTP = pad_to_mod(MT, 2, 32);
  static constexpr int NT = 256, CPP = 4, NP = KC / CPP;
  static constexpr int JX = (CPP * PER_CH + NT - 1) / NT;
  static constexpr int GX = 16 * (MT / 4) / NT;  // dY float4 items per thread
};

// WV waves per workgroup (4 or 8).  Waves 0-3 / 4-7 take the taps round-robin (tap = (wave & 3) + 4t); with 8 waves the two
// groups split the x-rows of the tile between them and write separate slabs.  8 waves share one staged tile, so the
// staging registers per thread halve and 16 waves per CU (4 per SIMD, <= 128 registers) hide the LDS latency that keeps a
// 2-waves-per-SIMD launch at 63 % matrix-pipe utilisation (PMC, profiles/).
// NCB = 16-cout blocks per workgroup (1..3): every staged x value (the B operand, one LDS read per tap) then feeds NCB MFMAs,
// as the forward kernel does with NBT — (4 NCB + 28) LDS dwords per 28 NCB MFMAs, and the x halo tile is re-staged NCB times
// less often.
// TGN = tap groups (4 or 8): with 8 waves and TGN = 8 every wave owns 3-4 taps of ALL rows instead of 7 taps of half the rows
// (12 accumulator registers less per cout block; with one cout block the kernel then fits 124 registers, i.e. four waves per SIMD,
// without spilling: 79 vs 74 TFLOP/s on 16->16 @96^3).  Interleaving the k-steps of two taps to break the dependent-accumulator
// chains (40 instead of 32 cycles) was measured slower: the other waves of the SIMD already fill those gaps.
template <class TC, int WV, int NCB, int TGN = 4>
__global__ __launch_bounds__(64 * WV, WV / 2) void conv3d_mfma_wgrad_static_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                              float* __restrict__ gwp, ConvGeom g) {
  constexpr int T = 27, NTW = (27 + TGN - 1) / TGN, TG = TGN, NT = 64 * WV, HALVES = WV / TGN, MB = 16 * NCB;
  constexpr int PS = TC::PS, PXL = TC::PXL, PY = TC::PY, PZ = TC::PZ, MT = TC::MT, MTP = TC::MTP;
  constexpr int NP = TC::NP, CPP = TC::CPP, Q = TC::Q;
  constexpr int JX = (CPP * TC::PER_CH + NT - 1) / NT;
  constexpr int GX = (MB * (MT / 4) + NT - 1) / NT;
  constexpr int ROWS = MT / 16, RPH = ROWS / HALVES;
  static_assert(ROWS % HALVES == 0, "tile rows must split evenly over the wave groups");
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
  const int tg = wid % TGN, half = wid / TGN;
  const int lq = lane >> 4, lr = lane & 15;
  const int ntiles = g.ntz * g.nty * g.ntx;

  // tile-invariant parts of the staging tables (row / column of each item inside the halo tile)
  int xpz[JX], xpy[JX], xq[JX], xc[JX], xl[JX];
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int it = threadIdx.x + j * NT;
    xl[j] = -1; xpz[j] = 0; xpy[j] = 0; xq[j] = 0; xc[j] = 0;
    if (it < CPP * TC::PER_CH) {
      const int q = it % Q;
      int r = it / Q;
      const int py = r % PY;
      r /= PY;
      xpz[j] = r % PZ; xpy[j] = py; xq[j] = q; xc[j] = r / PZ;
      xl[j] = xc[j] * PS + (xpz[j] * PY + py) * PXL + q * 4;
    }
  }
  int gco[GX], gvt[GX];
#pragma unroll
  for (int i = 0; i < GX; ++i) {
    const int it = threadIdx.x + i * NT;
    gvt[i] = (it % (MT / 4)) * 4;
    gco[i] = it < MB * (MT / 4) ? it / (MT / 4) : -1;
  }
  float4 xv[NP][JX];
  float4 gv[GX];
  auto load_tile = [&](int bt) {
    const int x0 = (bt % g.ntx) * TC::TX;
    const int y0 = ((bt / g.ntx) % g.nty) * TC::TY;
    const int z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int gx = x0 - TC::HX + xq[j] * 4, gyy = y0 - 1 + xpy[j], gz = z0 - 1 + xpz[j];
      const bool ok = xl[j] >= 0 && gz >= 0 && gz < g.D && gyy >= 0 && gyy < g.H && gx >= 0 && gx + 3 < g.W;
      const long off = gz * HW + (long)gyy * g.W + gx;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int ch = c0 + p * CPP + xc[j];
        xv[p][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && ch < g.Cin) xv[p][j] = *reinterpret_cast<const float4*>(xb + (long)ch * DHW + off);
      }
    }
#pragma unroll
    for (int i = 0; i < GX; ++i) {
      const int vt = gvt[i];
      const int tx = vt % TC::TX, t2 = vt / TC::TX;
      const int gz = z0 + t2 / TC::TY, gyy = y0 + t2 % TC::TY, gx = x0 + tx;
      gv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gco[i] >= 0 && co0 + gco[i] < g.Cout && gz < g.D && gyy < g.H && gx + 3 < g.W)
        gv[i] = *reinterpret_cast<const float4*>(gb + (long)(co0 + gco[i]) * DHW + gz * HW + (long)gyy * g.W + gx);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < JX; ++j)
        if (xl[j] >= 0) {  // PS is even, not a multiple of 4: 8-byte stores
          float* d = Xs + p * CPP * PS + xl[j];
          *reinterpret_cast<float2*>(d) = make_float2(xv[p][j].x, xv[p][j].y);
          *reinterpret_cast<float2*>(d + 2) = make_float2(xv[p][j].z, xv[p][j].w);
        }
#pragma unroll
    for (int i = 0; i < GX; ++i) {
      if (gco[i] < 0) continue;
      float* d = Gs + gco[i] * MTP + gvt[i];
      *reinterpret_cast<float2*>(d) = make_float2(gv[i].x, gv[i].y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(gv[i].z, gv[i].w);
    }
  };

  int bb[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int tap = tg + t * TG < T ? tg + t * TG : T - 1;   // a wave without a last tap prefetches (and ignores) tap 26
    const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
    bb[t] = lr * PS + lq + (TC::HX - 1) + (dz * PY + dy) * PXL + dx;
  }
  const int ab = lr * MTP + lq;
  f32x4 acc[NCB][NTW];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // each workgroup walks a contiguous run of tiles (x-neighbours back to back: their shared halo is still in L2), and with
  // g.remap the runs of one XCD are contiguous too
  const int lbx = g.remap ? xcd_chunked(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int bt = (int)((long)lbx * ntiles / gridDim.x);
  const int bt_end = (int)((long)(lbx + 1) * ntiles / gridDim.x);
  if (bt < bt_end) load_tile(bt);
  for (; bt < bt_end; ++bt) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (bt + 1 < bt_end) load_tile(bt + 1);
    if constexpr (NCB == 2) {
      // one x-row of the tile (16 voxels = 4 k-steps) at a time: per operand ONE row base register and the four k-steps at
      // dword offsets 0/4/8/12, which the compiler pairs into ds_read2_b32 (8-bit dword offsets).
      // Software pipeline (two cout blocks at two waves per SIMD: +3-8 % on the 32-multiple layers): the B
      // operand of the NEXT (row, tap) step — and, at the last tap of a row, the A operand of the next row — is read from LDS
      // before the MFMAs of the current step are issued (left to itself the compiler emits read, wait lgkmcnt(0), 4 MFMAs for
      // every half step, i.e. the whole LDS latency in front of every 128 cycles of matrix work).
      auto load_a = [&](int rr, float (&a)[NCB][4]) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          int ai = ab + cb * 16 * MTP + 16 * rr + half * (16 * RPH);
          ICL_OPAQUE_INT(ai);
#pragma unroll
          for (int s = 0; s < 4; ++s) a[cb][s] = Gs[ai + 4 * s];
        }
      };
      auto load_b = [&](int rr, int t, float (&bv)[4]) {
        // RPH is a multiple of TY for every tile used (4x4x16: 16 rows, TY 4; 8 rows per half = 2 z-planes)
        int bi = bb[t] + ((rr / TC::TY) * PY + (rr % TC::TY)) * PXL + half * ((RPH / TC::TY) * PY * PXL);
        ICL_OPAQUE_INT(bi);
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[s] = Xs[bi + 4 * s];
      };
      float a[NCB][4], an[NCB][4], b[4], bn[4];
      load_a(0, a);
      load_b(0, 0, b);
#pragma unroll
      for (int rr = 0; rr < RPH; ++rr) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (t + 1 < NTW) load_b(rr, t + 1, bn);
          else if (rr + 1 < RPH) { load_a(rr + 1, an); load_b(rr + 1, 0, bn); }
          ICL_SCHED_BARRIER();
          if (t < NTW - 1 || tg + t * TG < T) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb) acc[cb][t] = icl_mfma_16x16x4(a[cb][s], b[s], acc[cb][t]);
          }
          ICL_SCHED_BARRIER();
#pragma unroll
          for (int s = 0; s < 4; ++s) b[s] = bn[s];
          if (t == NTW - 1 && rr + 1 < RPH) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
              for (int s = 0; s < 4; ++s) a[cb][s] = an[cb][s];
          }
        }
      }
    } else {
      // one cout block (124 registers, four waves per SIMD): the explicit prefetch spills (128 registers + 40 B scratch) and is 1-3 %
      // slower — the other waves cover the LDS latency; three cout blocks: 256 registers + 164 B scratch with the prefetch, 6-8 %
      // slower (48->48 @96^3: 97 -> 89 TFLOP/s)
      // one x-row of the tile (16 voxels = 4 k-steps) at a time: per operand ONE row base register and the four k-steps at
      // dword offsets 0/4/8/12, which the compiler pairs into ds_read2_b32 (8-bit dword offsets) — half the LDS instructions
#pragma unroll
      for (int rr = 0; rr < RPH; ++rr) {
        // rows of this wave group; (row / TY, row % TY) must be compile-time per unrolled step, so the half enters as an offset
        float a[NCB][4];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          int ai = ab + cb * 16 * MTP + 16 * rr + half * (16 * RPH);
          ICL_OPAQUE_INT(ai);
#pragma unroll
          for (int s = 0; s < 4; ++s) a[cb][s] = Gs[ai + 4 * s];
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (t < NTW - 1 || tg + t * TG < T) {
            // RPH is a multiple of TY for every tile used (4x4x16: 16 rows, TY 4; 8 rows per half = 2 z-planes)
            int bi = bb[t] + ((rr / TC::TY) * PY + (rr % TC::TY)) * PXL + half * ((RPH / TC::TY) * PY * PXL);
            ICL_OPAQUE_INT(bi);
            float b[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) b[s] = Xs[bi + 4 * s];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb) acc[cb][t] = icl_mfma_16x16x4(a[cb][s], b[s], acc[cb][t]);
          }
        }
      }
    }
  }
  const int ci = c0 + lr;
  if (ci < g.CinP) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int tap = tg + t * TG;
      if (tap >= T) continue;
      // every (split, batch, wave group) owns one slab of packed partial sums: plain 16-byte stores, no atomics;
      // reduce_unpack_wgrad_kernel adds the slabs in a fixed order (bitwise reproducible gradients)
      const long slab = ((long)blockIdx.z * gridDim.x + blockIdx.x) * HALVES + half;
      float* dst = gwp + slab * ((long)T * g.CinP * g.CoutP) + ((long)tap * g.CinP + ci) * g.CoutP + co0 + lq * 4;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        if (co0 + cb * 16 < g.CoutP)
          *reinterpret_cast<float4*>(dst + cb * 16) = make_float4(acc[cb][t][0], acc[cb][t][1], acc[cb][t][2], acc[cb][t][3]);
    }
  }
}

}  // namespace icl
