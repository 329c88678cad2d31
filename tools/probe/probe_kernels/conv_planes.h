// conv_planes.h — the split-product 3x3x3 convolution (conv_bf16x3.h) fed by LDS-DMA from operand planes its producer wrote.
//
// conv_bf16x3.h stages an fp32 NCDHW halo tile through registers: 40 one-dword loads, 187 VALU of operand split and 15 ds_write_b128
// per thread and work item — by the in-kernel stamps of round 3, ≈8 k of the ≈18.5 k cycles of a work item, beside 10.75 k of matrix
// work.  Here the PRODUCER of the activation (normalise + ReLU, pooling, up-sampling, the normalisation's backward pass) writes, next
// to its fp32 result, the three bf16 terms of every value in the layout the multiply loop reads from LDS,
//     P[sample][chunk of 16 channels][q = 2 * split + half][D*H*W positions + 8 zero positions]   of uint4 = 8 packed bf16,
// and a halo tile is 102 LDS-DMA instructions per workgroup (buffer_load_dwordx4 ... lds: 64 positions x 16 bytes each, per-lane
// SOURCE address = the halo position's voxel, destination = the LDS image itself): no staging registers, no split VALU, no ds_write.
// Positions outside the volume and the pad positions of the LDS image read the zero positions behind the plane (always inside the
// descriptor: nothing depends on what the hardware does with a range-checked LDS-DMA lane).
// The halo tile is single-buffered (104 KB of the 160 KB; two do not fit): the DMA of the next work item is issued behind the
// barrier that ends the multiply phase and lands while the workgroup writes the outputs of this one.
// Multiply loop, LDS images, weight planes, tile order and the order of the floating-point sums are those of
// conv3d_bf16x3_fwd_kernel<NBT, 8, 60>: the results are bit-identical to it.
// Reference op: nn.Conv3d(k=3, pad=1) inside UnetConv3 (/root/reference/code/networks/utils.py:104,107) and its input gradient.
#pragma once

namespace icl {

// zero positions behind every plane (128 bytes: planes stay 128-byte aligned)
constexpr int kPlanePad = 8;
__host__ __device__ inline long planes_pitch(long dhw) { return dhw + kPlanePad; }

struct Bf3PGeom {
  int Cout, CoutP;            // CoutP: extent of the split weights (conv_bf16x3_split_weights_kernel)
  int D, H, W;
  int ntz, nty, ntx, ntiles;  // tiles per sample, ntiles = batch * ntz * nty * ntx
  int nchunks;                // input chunks of 16 channels of THIS convolution (the planes pointer addresses its first chunk)
  long ppitch;                // planes_pitch(D*H*W), uint4 units
  long p_bstride;             // uint4 units between samples of the planes tensor (all chunks of the buffer x 6 x ppitch)
  long y_bstride;
  int dbg;                    // timing ablations (probe builds; results are wrong when set): 1 no halo DMA after the first item, 2 no output stores
};

// fp32 [N][C][S] (sample stride x_bstride) -> planes of chunks chunk0 .. of a planes tensor with p_bstride uint4 per sample.
// The stand-alone producer: used where no fused producer exists (and by the probes); C % 16 == 0.
__global__ __launch_bounds__(256) void planes_from_f32_kernel(const float* __restrict__ x, uint4* __restrict__ planes, int N, int C, long S,
                                                              long x_bstride, long p_bstride) {
  const long pitch = planes_pitch(S);
  const long total = (long)N * (C / 8) * pitch;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long pos = it % pitch, no = it / pitch;
    const int oct = (int)(no % (C / 8)), n = (int)(no / (C / 8));
    uint4* d = planes + (long)n * p_bstride + ((long)(oct >> 1) * 6 + (oct & 1)) * pitch + pos;
    if (pos >= S) {      // the zero positions
      d[0] = d[2 * pitch] = d[4 * pitch] = make_uint4(0u, 0u, 0u, 0u);
      continue;
    }
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = x[(long)n * x_bstride + (long)(oct * 8 + c) * S + pos];
    uint4 o1, o2, o3;
    bf3_split8(v, o1, o2, o3);
    d[0] = o1; d[2 * pitch] = o2; d[4 * pitch] = o3;
  }
}

// Configurations (NBT cout blocks of 16, TY rows of y per tile, NWV waves, DBUF):
//   <1, 8, 8, false>  one workgroup per CU, 4 x 8 x 16 tile, the three weight planes of a chunk resident (by DMA): the halo DMA of the
//                     next item is issued behind the multiply phase and lands beside the epilogue only — measured 0.91-0.97x of the
//                     shipped kernel on the 96^3 layers (profiles/r4_planes_dma_v1_probe.txt): the landing time is exposed.
//   <1, 4, 4, false>  4 x 4 x 16 tile, four waves, 77 KB: TWO workgroups per CU — while one waits for its halo tile the other multiplies.
//   <2, 4, 8, true>   4 x 4 x 16 tile, eight waves of two row blocks, the halo tile DOUBLE-buffered (2 x 62 KB + one 30 KB weight plane):
//                     the DMA of item i + 1 is issued at the top of item i.
//   <2|3, 8, 8, false> as the shipped kernel with the halo by DMA (weights one dz plane at a time through registers).
// FLAT: the 2 x 8 x 24 tile of the 24^3 level (eight waves of three row blocks).
template <int NBT, int TY, int NWV, bool DBUF, bool FLAT = false>
struct PlanesCfg {
  typedef typename std::conditional<FLAT, Bf3F24, Bf3T<TY>>::type TC;
  static constexpr int MB = FLAT ? 3 : 4 * TY / NWV;                    // row blocks of 16 outputs per wave
  static constexpr int NT = 64 * NWV;
  static constexpr bool WHOLE = NBT == 1 && TY == 8 && !DBUF && !FLAT;  // three weight planes resident, filled by DMA
  static constexpr int NPOSP = (TC::NPOS + 1 + 15) / 16 * 16;           // LDS plane pitch (a multiple of 16: the halves stay 256 B-aligned apart)
  static constexpr int NBLK = (NPOSP + 63) / 64;                        // DMA blocks of 64 positions per plane (the last one may be partial)
  static constexpr int KT = (NBLK + NWV - 1) / NWV;                     // position blocks per wave
  static constexpr int XBUF = 6 * NPOSP;                                // uint4 per halo buffer
  static constexpr int WPL = WHOLE ? 3 : 1;
  static constexpr int WS_U4 = WPL * 6 * Bf3::SLOTS * 16 * NBT;
  static constexpr size_t LDS_BYTES = (size_t)((DBUF ? 2 : 1) * XBUF + WS_U4) * 16;
  static_assert(MB * NWV * 16 == TC::TZ * TC::TY * TC::TX, "the waves cover the tile");
};

#if defined(PLANES_STAMPS)
// in-kernel stamps (probe builds): waves 0 and NWV / 2 of workgroup 0 record s_memtime at the phase boundaries of work items 2..4
__device__ long long g_planes_stamps[2 * 3 * 16];
#define PL_STAMP(k)                                                                                                   \
  do {                                                                                                                \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (wid == 0 || wid == NWV / 2) && lane == 0 && item_no >= 2 && item_no < 5) \
      g_planes_stamps[((wid ? 1 : 0) * 3 + item_no - 2) * 16 + (k)] = clock64();                                      \
  } while (0)
#else
#define PL_STAMP(k) ((void)0)
#endif

template <int NBT, int TY, int NWV, bool DBUF, bool FLAT = false>
__global__ __launch_bounds__(64 * NWV) void conv3d_planes_fwd_kernel(const uint4* __restrict__ planes, const uint4* __restrict__ wsplit,
                                                                     const float* __restrict__ bias, float* __restrict__ y, Bf3PGeom g) {
  typedef PlanesCfg<NBT, TY, NWV, DBUF, FLAT> C;
  typedef typename C::TC TC;
  constexpr int MB = C::MB, NT = C::NT, NPOSP = C::NPOSP, NBLK = C::NBLK, KT = C::KT, XBUF = C::XBUF;
  constexpr bool WHOLE = C::WHOLE, PIPE_B2 = NBT < 3;
  constexpr int NB = 16 * NBT, PX = TC::PX, PY = TC::PY;
  constexpr int WITEMS = C::WS_U4, WU = (WITEMS + NT - 1) / NT;
  static_assert(!WHOLE || WITEMS % 64 == 0, "whole DMA blocks");
  // the ragged last round of position blocks (block NBLK - 1 alone): dealt over the waves by plane instead of all six to wave 0
  constexpr bool LAST_BY_PLANE = NBLK - NWV * (KT - 1) == 1 && NWV >= 6;
  ICL_DYN_LDS(uint4, lds);
  uint4* Xs = lds;
  uint4* Ws = lds + (DBUF ? 2 : 1) * XBUF;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  int wid = tid >> 6;
  ICL_WAVE_UNIFORM(wid);
  const int half = lq & 1, tp = lq >> 1;
  const int n0 = blockIdx.y * NB;
  const int HW = g.H * g.W;
  const long DHW = (long)g.D * HW;
  const int tiles_per = g.ntz * g.nty * g.ntx;

  // ---- halo DMA: wave w moves the position blocks w, w + NWV, .. of all six planes.  Tile-invariant part of a lane's source offset:
  // its halo position relative to the tile origin; t_zyx < 0: a pad position of the LDS image (reads the plane's zero positions),
  // t_zyx == -2: beyond the image (the lane takes no part: a partial last block)
  int t_rel[KT], t_zyx[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const int pbk = (LAST_BY_PLANE && k == KT - 1) ? NBLK - 1 : wid + NWV * k;
    const int pos = pbk * 64 + lane;
    const int px = pos % PX, row = pos / PX, py = row % PY, pz = row / PY;
    t_zyx[k] = pos < TC::NPOS ? (pz << 16) | (py << 8) | px : pos < NPOSP ? -1 : -2;
    t_rel[k] = (pz - 1) * HW + (py - 1) * g.W + (px - 1);
  }
  auto issue_x = [&](int tile, int chunk, uint4* dst) {
    const int b = tile / tiles_per, bt = tile % tiles_per;
    const int x0 = (bt % g.ntx) * TC::TX, y0 = ((bt / g.ntx) % g.nty) * TC::TY, z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
    const icl_rsrc_t xr = icl_make_rsrc(planes + (long)b * g.p_bstride + (long)chunk * 6 * g.ppitch, (unsigned)(6 * g.ppitch * 16));
    const int toff = z0 * HW + y0 * g.W + x0;
    const unsigned pb = (unsigned)g.ppitch * 16u;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int gz = z0 - 1 + (t_zyx[k] >> 16), gy = y0 - 1 + ((t_zyx[k] >> 8) & 255), gx = x0 - 1 + (t_zyx[k] & 255);
      const bool ok = (t_zyx[k] >= 0) & ((unsigned)gz < (unsigned)g.D) & ((unsigned)gy < (unsigned)g.H) & ((unsigned)gx < (unsigned)g.W);
      const unsigned vo = ok ? (unsigned)(t_rel[k] + toff) * 16u : (unsigned)DHW * 16u;      // outside: the plane's zero positions
      if (LAST_BY_PLANE && k == KT - 1) {
        if (wid < 6 && t_zyx[k] != -2) icl_buffer_load_lds_b128(xr, dst + wid * NPOSP + (NBLK - 1) * 64, vo, wid * pb);
      } else if (wid + NWV * k < NBLK) {
        const bool full = (wid + NWV * k + 1) * 64 <= NPOSP;      // wave-uniform; a partial block masks its lanes beyond the image
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          if (full) icl_buffer_load_lds_b128(xr, dst + q * NPOSP + (wid + NWV * k) * 64, vo, q * pb);
          else if (t_zyx[k] != -2) icl_buffer_load_lds_b128(xr, dst + q * NPOSP + (wid + NWV * k) * 64, vo, q * pb);
        }
      }
    }
  };
  // weights (already split: conv_bf16x3_split_weights_kernel).  WHOLE: the three dz planes of a chunk (45 blocks of 64 slots) by DMA;
  // else one dz plane at a time through registers, as in conv3d_bf16x3_fwd_kernel
  auto issue_w = [&](int chunk) {
    const uint4* src = wsplit + (long)chunk * 3 * 6 * Bf3::SLOTS * g.CoutP + n0;
    const icl_rsrc_t wr = icl_make_rsrc(src, (unsigned)((3 * 6 * Bf3::SLOTS * g.CoutP - n0) * 16));
#pragma unroll
    for (int i = 0; i < (WITEMS / 64 + NWV - 1) / NWV; ++i) {
      const int blk = wid + NWV * i;
      if (blk < WITEMS / 64) icl_buffer_load_lds_b128(wr, Ws + blk * 64, (unsigned)((blk * 4 + (lane >> 4)) * g.CoutP + (lane & 15)) * 16u, 0u);
    }
  };
  uint4 wv[WHOLE ? 1 : WU];
  auto load_w = [&](int chunk, int dz) {
    const uint4* src = wsplit + (long)(chunk * 3 + dz) * 6 * Bf3::SLOTS * g.CoutP + n0;
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      const int it = tid + i * NT;
      wv[i] = make_uint4(0u, 0u, 0u, 0u);
      if (it < WITEMS && n0 + it % NB < g.CoutP) wv[i] = src[(long)(it / NB) * g.CoutP + it % NB];
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      const int it = tid + i * NT;
      if (it < WITEMS) Ws[it] = wv[i];
    }
  };

  // ---- operand bases: wave w owns the (z, y) rows MB w .. MB w + MB - 1 of the tile (FLAT: row blocks 3 w .. 3 w + 2 of the flattened tile)
  const int wz = (MB * wid) / TC::TY, wy = (MB * wid) % TC::TY;
  int moff[MB];
  int lanepos;
  if (FLAT) {
    int off[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const int p = 16 * (MB * wid + m) + lr, pz = p / (TC::TY * TC::TX), py = (p / TC::TX) % TC::TY, px = p % TC::TX;
      off[m] = (pz * PY + py) * PX + px;
    }
    lanepos = off[0];
#pragma unroll
    for (int m = 0; m < MB; ++m) moff[m] = off[m] - off[0];
  } else {
    lanepos = (wz * PY + wy) * PX + lr;
#pragma unroll
    for (int m = 0; m < MB; ++m) moff[m] = m * PX;
  }
  const uint4* xa0 = Xs + half * NPOSP + lanepos;
  const uint4* wb = Ws + (half * Bf3::SLOTS + tp) * NB + lr;

  // the bias is fetched (and waited for) before the first DMA is issued: an ordinary load whose result is first used while LDS-DMA
  // is in flight makes hipcc wait vmcnt(0) at that use — in front of every output store of the epilogue, which is exactly where the
  // next halo tile is supposed to land unattended
  float bv[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int co = n0 + j * 16 + lr;
    bv[j] = (bias && co < g.Cout) ? bias[co] : 0.f;
    ICL_PIN1(bv[j]);
  }
  f32x4 acc[MB][NBT];
  uint4 pa1[MB], pa23[MB][2], pb[PIPE_B2 ? 2 : 1][3][NBT];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < NBT; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // tile order: every XCD walks its own contiguous eighth of the tile list (conv_bf16x3.h); gridDim.x % 8 == 0
  const int per_xcd = (g.ntiles + 7) / 8, xcd = blockIdx.x & 7, wgs_per_xcd = gridDim.x >> 3;
  const int xcd_end = (xcd + 1) * per_xcd < g.ntiles ? (xcd + 1) * per_xcd : g.ntiles;
  auto next_tile = [&](int t) { return t + wgs_per_xcd < xcd_end ? t + wgs_per_xcd : g.ntiles; };
  int tile = xcd * per_xcd + (blockIdx.x >> 3), chunk = 0;
  if (tile >= xcd_end) tile = g.ntiles;
  if (tile < g.ntiles) {
    if (WHOLE) issue_w(0); else load_w(0, 0);
    issue_x(tile, 0, Xs);
  }
  int buf = 0;                                 // DBUF: the halo buffer of the current item
  int item_no = -1;
  (void)item_no;

  while (tile < g.ntiles) {
    int ntile = tile, nchunk = chunk + 1;
    if (nchunk == g.nchunks) { nchunk = 0; ntile = next_tile(tile); }
    ++item_no;
    PL_STAMP(0);
    const uint4* xa = xa0 + (DBUF && buf ? XBUF : 0);
    auto frag_ptr = [&](int sdz, int spair) {
      const int tA = 10 * sdz + 2 * spair, tB = tA + 1 < 27 ? tA + 1 : 26;
      const int offA = (tA / 9) * PY * PX + ((tA / 3) % 3) * PX + tA % 3, offB = (tB / 9) * PY * PX + ((tB / 3) % 3) * PX + tB % 3;
      return xa + (tp ? offB : offA);
    };
    auto load_b = [&](int bi, int sdz, int spair, int s0 = 0, int s1 = 3) {
#pragma unroll
      for (int s = s0; s < s1; ++s)
#pragma unroll
        for (int j = 0; j < NBT; ++j)
          pb[bi][s][j] = wb[((WHOLE ? sdz * 6 : 0) * Bf3::SLOTS + s * 2 * Bf3::SLOTS + spair * 2) * NB + j * 16];
    };
    auto load_x1 = [&](int sdz, int spair) {
      const uint4* xp = frag_ptr(sdz, spair);
#pragma unroll
      for (int m = 0; m < MB; ++m) pa1[m] = xp[moff[m]];
    };
    auto load_x23 = [&](int sdz, int spair) {
      const uint4* xp = frag_ptr(sdz, spair);
#pragma unroll
      for (int m = 0; m < MB; ++m) pa23[m][1] = xp[4 * NPOSP + moff[m]];      // a3 first: its products lead the Y half
#pragma unroll
      for (int m = 0; m < MB; ++m) pa23[m][0] = xp[2 * NPOSP + moff[m]];
    };
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      if (WHOLE) {
        if (dz == 0) {
          ICL_WAIT_VMEM();                     // this wave's share of the halo tile (and weights) has landed ...
          PL_STAMP(1);
          __syncthreads();                     // ... and everybody else's
          PL_STAMP(2);
        }
      } else {
        // dz = 0: the barrier that ended the last item (!DBUF) / the one below (DBUF) has retired every read of the weight plane
        if (dz > 0 || DBUF) __syncthreads();
        store_w();
        if (dz == 0) ICL_WAIT_VMEM();          // this wave's share of the halo tile has landed
        if (dz == 0) PL_STAMP(1);
        __syncthreads();
        if (dz == 0) PL_STAMP(2);
        if (dz < 2) load_w(chunk, dz + 1);
        else if (ntile < g.ntiles) load_w(nchunk, 0);
        // DBUF: the other buffer was last read by the previous item, which everybody has left: fill it with the next item now
        if (DBUF && dz == 0 && ntile < g.ntiles && !(g.dbg & 1)) issue_x(ntile, nchunk, Xs + (buf ? 0 : XBUF));
      }
      // A pair's 24 NBT products in two halves: X = the a1 terms (a1 b3, a1 b2, a1 b1), Y = (a3 b1, a2 b2, a2 b1); the LDS reads of a half
      // are issued behind the MFMAs of the half before it (conv_bf16x3.h, variant 60)
      const int np = dz < 2 ? 5 : 4;
      if (!WHOLE || dz == 0) {
        load_b(PIPE_B2 ? (5 * dz) & 1 : 0, dz, 0);
        load_x1(dz, 0);
      }
#pragma unroll
      for (int pair = 0; pair < np; ++pair) {
        const int cur = PIPE_B2 ? (5 * dz + pair) & 1 : 0;
        load_x23(dz, pair);
#pragma unroll
        for (int sb = 2; sb >= 0; --sb)
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int j = 0; j < NBT; ++j) acc[m][j] = icl_mfma_16x16x32_bf16(pa1[m], pb[cur][sb][j], acc[m][j]);
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
          ICL_SCHED_GROUP(0x008, NBT);
          ICL_SCHED_GROUP(0x100, 1);
        }
        ICL_SCHED_GROUP(0x008, MB * NBT);
        ICL_SCHED_BARRIER();
        const bool more = pair + 1 < np || (WHOLE && dz < 2);
        const int ndz = pair + 1 < np ? dz : dz + 1, npair = pair + 1 < np ? pair + 1 : 0;
        if (more) {
          if (PIPE_B2) load_b(cur ^ 1, ndz, npair, 2, 3);
          load_x1(ndz, npair);
          if (PIPE_B2) load_b(cur ^ 1, ndz, npair, 0, 2);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          constexpr int sa[3] = {1, 0, 0}, sbb[3] = {0, 1, 0};      // a3 b1, a2 b2, a2 b1
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int j = 0; j < NBT; ++j) acc[m][j] = icl_mfma_16x16x32_bf16(pa23[m][sa[t]], pb[cur][sbb[t]][j], acc[m][j]);
        }
        if (more) {
          constexpr int R = MB + (PIPE_B2 ? 3 * NBT : 0);      // reads of this half-step
          constexpr int NM = 3 * MB * NBT;                     // its MFMAs
#pragma unroll
          for (int i = 0; i < (R < NM ? R : NM); ++i) {
            ICL_SCHED_GROUP(0x008, 1);
            ICL_SCHED_GROUP(0x100, 1);
          }
          if (NM > R) ICL_SCHED_GROUP(0x008, NM - R);
        }
        ICL_SCHED_BARRIER();
        if (more && !PIPE_B2) load_b(0, ndz, npair);
      }
      PL_STAMP(3 + dz);
    }
    if (!DBUF) {
      __syncthreads();                         // everyone has finished reading the halo tile and the weights
      PL_STAMP(6);
      if (ntile < g.ntiles && !(g.dbg & 1)) {
        issue_x(ntile, nchunk, Xs);            // lands while the outputs below are written
        if (WHOLE && g.nchunks > 1) issue_w(nchunk);
      }
    }
    PL_STAMP(7);
    if (chunk == g.nchunks - 1 && !(g.dbg & 2)) {
      // ---- epilogue: lane holds x = 4 lq + r of row (wid, m), column co = n0 + 16 j + lr
      const int b = tile / tiles_per, bt = tile % tiles_per;
      const int x0 = (bt % g.ntx) * TC::TX, y0 = ((bt / g.ntx) % g.nty) * TC::TY, z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
      float* yb = y + (long)b * g.y_bstride;
#pragma unroll
      for (int j = 0; j < NBT; ++j) {
        const int co = n0 + j * 16 + lr;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          int gz, gy, gx;
          if (FLAT) {
            const int p = 16 * (MB * wid + m) + 4 * lq;
            gz = z0 + p / (TC::TY * TC::TX); gy = y0 + (p / TC::TX) % TC::TY; gx = x0 + p % TC::TX;
          } else {
            gz = z0 + wz; gy = y0 + wy + m; gx = x0 + 4 * lq;
          }
          if (co < g.Cout && gz < g.D && gy < g.H && gx < g.W)
            *reinterpret_cast<float4*>(yb + (long)co * DHW + (long)gz * HW + (long)gy * g.W + gx) =
                make_float4(acc[m][j][0] + bv[j], acc[m][j][1] + bv[j], acc[m][j][2] + bv[j], acc[m][j][3] + bv[j]);
          acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    PL_STAMP(8);
    tile = ntile;
    chunk = nchunk;
    buf ^= 1;
  }
}

}  // namespace icl
