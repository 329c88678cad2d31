// conv_bf16x3_ws.h — the split-product 3x3x3 convolution (conv_bf16x3.h) with dedicated loader waves (round 4).
//
// What bounds conv3d_bf16x3_fwd_kernel<., 8, 60> (in-kernel stamps of round 4 on the LDS-DMA probe kernels, profiles/r4_planes_dma.md):
// the halo tile of a work item is 69 KB of fp32 (104 KB as bf16 planes) and a CU takes in ~14 B per clock while every CU fetches at
// once (the Infinity-Cache / HBM rate of the chip, 7.5 TB/s over 256 CUs) — 5 k of a 20 k-cycle item, next to 10.75 k cycles of matrix
// work.  A wave that issues those loads between its own MFMAs stalls in the vector-memory queue and its in-order MFMA stream stalls
// with it; a single-buffered LDS-DMA of the next tile can only start when the multiply phase has ended.  The tile in flight has to sit
// somewhere that is neither the LDS tile being multiplied (two do not fit) nor the registers of a wave that multiplies:
//   waves 0..7   consumers: the multiply loop of variant 60 (fragments woven between the MFMAs) and the epilogue — no loads of x;
//   waves 8..11  loaders: fetch the fp32 halo tile of the NEXT work item (buffer loads, zero fill by the descriptor), split it into the
//                three bf16 planes in registers (108 VGPRs per lane) while the consumers multiply, and write the LDS image between
//                the two barriers of the item (27 ds_write_b128 per lane, no VALU).
// Same LDS images, same weight planes, same tile order, same order of the floating-point sums as conv3d_bf16x3_fwd_kernel: results are
// bit-identical to it.  Three waves per SIMD: every wave must fit 168 registers.
// Reference op: nn.Conv3d(k=3, pad=1) inside UnetConv3 (/root/reference/code/networks/utils.py:104,107) and its input gradient.
#pragma once

namespace icl {

#if defined(WS_STAMPS)
// in-kernel stamps (probe builds): consumer waves 0 and 4 (one SIMD) and loader wave 8 of workgroup 0, work items 2..4
__device__ long long g_ws_stamps[3 * 3 * 16];
#define WS_STAMP(k)                                                                                                      \
  do {                                                                                                                   \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (wid == 0 || wid == 4 || wid == 8) && lane == 0 && item_no >= 2 && item_no < 5) \
      g_ws_stamps[((wid >> 2) * 3 + item_no - 2) * 16 + (k)] = clock64();                                                \
  } while (0)
#else
#define WS_STAMP(k) ((void)0)
#endif

template <int NBT, bool FLAT = false, bool LF = false>
__global__ __launch_bounds__(768) void conv3d_bf16x3_fwd_ws_kernel(const float* __restrict__ x, const uint4* __restrict__ wsplit,
                                                                   const float* __restrict__ bias, float* __restrict__ y, Bf3Geom g) {
  typedef typename std::conditional<FLAT, Bf3F24, Bf3T<8>>::type TC;
  constexpr int MB = FLAT ? 3 : 4;                      // row blocks per consumer wave
  constexpr bool WHOLE = NBT == 1, PIPE_B2 = NBT < 3;
  constexpr int WPL = WHOLE ? 3 : 1;
  constexpr int NB = 16 * NBT, PX = TC::PX, PY = TC::PY, NPOSP = TC::NPOSP;
  constexpr int NC = 512, NL = 256;                     // consumer / loader threads
  constexpr int WITEMS = WPL * 6 * Bf3::SLOTS * NB, WU = (WITEMS + NC - 1) / NC;
  constexpr int ITEMS = 2 * TC::NPOS, ROUNDS = (ITEMS + NL - 1) / NL;      // loader staging rounds: item = (channel octet, halo position)
  ICL_DYN_LDS(uint4, lds);
  uint4* Xs = lds;
  uint4* Ws = lds + TC::XS_U4;
  // LF ("loaders first", round 5 experiment): the four loader waves are the workgroup's OLDEST waves (hardware wave ids 0..3) instead of its
  // youngest — the SIMD's issue arbiter prefers the older wave at equal priority.  Roles are expressed through `wid` / `tid` as the code
  // below has always used them: consumers 0..7 (threads 0..511), loaders 8..11 (threads 512..767).
  const int htid = threadIdx.x;
  const int tid = LF ? (htid < NL ? htid + NC : htid - NL) : htid;
  const int lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  int wid = tid >> 6;
  ICL_WAVE_UNIFORM(wid);
  const int n0 = blockIdx.y * NB;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const int tiles_per = g.ntz * g.nty * g.ntx;

  // zero the pad positions once (read by the zero slot: garbage * 0 must not be NaN)
  for (int i = tid; i < 6 * (NPOSP - TC::NPOS); i += NC + NL)
    Xs[(i / (NPOSP - TC::NPOS)) * NPOSP + TC::NPOS + (i % (NPOSP - TC::NPOS))] = make_uint4(0u, 0u, 0u, 0u);

  // tile order: every XCD walks its own contiguous eighth of the tile list (conv_bf16x3.h); gridDim.x % 8 == 0
  const int per_xcd = (g.ntiles + 7) / 8, xcd = blockIdx.x & 7, wgs_per_xcd = gridDim.x >> 3;
  const int xcd_end = (xcd + 1) * per_xcd < g.ntiles ? (xcd + 1) * per_xcd : g.ntiles;
  auto next_tile = [&](int t) { return t + wgs_per_xcd < xcd_end ? t + wgs_per_xcd : g.ntiles; };
  int tile = xcd * per_xcd + (blockIdx.x >> 3), chunk = 0;
  if (tile >= xcd_end) tile = g.ntiles;

  if (wid >= 8) {
    // ================================================================================================ loader waves
    const int lt = tid - NC;
#if !defined(WS_LOADER_PRIO)
#define WS_LOADER_PRIO 0
#endif
    // the hardware issues the oldest wave of a SIMD first and the loaders are its youngest: left at equal priority a loader wave got
    // one instruction in per ~29 cycles (stamps: 20 k cycles to fetch one tile) behind two consumers that keep the issue port busy
    ICL_SETPRIO(WS_LOADER_PRIO);
    int s_zd[ROUNDS], s_rel[ROUNDS];                     // (pz, py, px, LDS slot) packed; source offset relative to the tile origin
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int it = lt + r * NL;
      const int o = it / TC::NPOS, pos = it % TC::NPOS;
      const int px = pos % PX, row = pos / PX, py = row % PY, pz = row / PY;
      s_zd[r] = it < ITEMS ? (pz << 27) | (py << 22) | (px << 16) | (o * NPOSP + pos) : -1;
      s_rel[r] = o * 8 * (int)DHW + (pz - 1) * (int)HW + (py - 1) * g.W + (px - 1);
    }
    uint4 pl[ROUNDS][3];                                 // the three packed planes of every staging item of the tile in flight
    // ALL loads of the next tile are issued at once, right behind barrier A (72 one-dword gathers per lane = 69 KB in flight per CU),
    // and split in issue order as they arrive: under chip-wide load a fetch comes back after ~3 us (tools/probe/halo_fetch_probe.hip:
    // 2.9 us per tile with 80 KB in flight and nothing else running), so a loader that keeps only three rounds (24 KB) in flight
    // needs 20 k cycles per tile (first version of this kernel: stamps in profiles/r4_ws_stamps.txt) — a latency bound, not an
    // issue or bandwidth bound.  Registers: 72 raw values up front, 12 per split round replace 8: 108 at the end.
    // (the raw values of a round land in the first two of its three plane registers and are split in place: a separate raw array
    // made hipcc keep 72 + 108 registers live and spill 91)
    icl_rsrc_t xr = icl_make_rsrc(x, 0u);
    int toff = 0, oz = 0, oy = 0, ox = 0;
    auto origin = [&](int t, int ch) {
      const int b = t / tiles_per, bt = t % tiles_per;
      ox = (bt % g.ntx) * TC::TX; oy = ((bt / g.ntx) % g.nty) * TC::TY; oz = (bt / (g.ntx * g.nty)) * TC::TZ;
      xr = icl_make_rsrc(x + (long)b * g.x_bstride + (long)ch * 16 * DHW, (unsigned)(16 * DHW * 4));
      toff = oz * (int)HW + oy * g.W + ox;
    };
    auto issue_all = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int gz = oz - 1 + ((s_zd[r] >> 27) & 15), gy = oy - 1 + ((s_zd[r] >> 22) & 31), gx = ox - 1 + ((s_zd[r] >> 16) & 63);
        const bool ok = (s_zd[r] >= 0) & ((unsigned)gz < (unsigned)g.D) & ((unsigned)gy < (unsigned)g.H) & ((unsigned)gx < (unsigned)g.W);
        const unsigned boff = ok ? (unsigned)(s_rel[r] + toff) * 4u : 0x80000000u;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = icl_buffer_load_f32(xr, boff, (unsigned)c * (unsigned)DHW * 4u);
        pl[r][0] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
        pl[r][1] = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
      }
    };
    auto split_range = [&](auto R0, auto R1) __attribute__((always_inline)) {
#pragma unroll
      for (int r = decltype(R0)::value; r < decltype(R1)::value; ++r)
        if (r < ROUNDS) {
          const float v[8] = {__uint_as_float(pl[r][0].x), __uint_as_float(pl[r][0].y), __uint_as_float(pl[r][0].z), __uint_as_float(pl[r][0].w),
                              __uint_as_float(pl[r][1].x), __uint_as_float(pl[r][1].y), __uint_as_float(pl[r][1].z), __uint_as_float(pl[r][1].w)};
          bf3_split8(v, pl[r][0], pl[r][1], pl[r][2]);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, ROUNDS / 3> I1;
    typedef std::integral_constant<int, ROUNDS> I2;
    auto deposit = [&]() {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        if (s_zd[r] < 0) continue;
        uint4* d = Xs + (s_zd[r] & 0xffff);
        d[0] = pl[r][0];
        d[2 * NPOSP] = pl[r][1];
        d[4 * NPOSP] = pl[r][2];
      }
    };
    if (tile < g.ntiles) {
      origin(tile, 0);
      issue_all();
      split_range(I0(), I2());
      deposit();
    }
    int item_no = -1;
    (void)item_no;
    while (tile < g.ntiles) {
      int ntile = tile, nchunk = chunk + 1;
      if (nchunk == g.nchunks) { nchunk = 0; ntile = next_tile(tile); }
      const bool more = ntile < g.ntiles;
      ++item_no;
      WS_STAMP(0);
      __syncthreads();                         // (A) the LDS image of this item (and the consumers' weights) is complete
      WS_STAMP(1);
      if (more) { origin(ntile, nchunk); issue_all(); }
      WS_STAMP(2);
      // more than one cout block: the consumers' barriers around their weight planes of dz = 1 and dz = 2, which the loaders take
      // part in: nothing that waits for memory in front of the first pair
      if (!WHOLE) { __syncthreads(); __syncthreads(); }
      if (more) split_range(I0(), I1());
      if (!WHOLE) { __syncthreads(); __syncthreads(); }
      WS_STAMP(3);
      if (more) split_range(I1(), I2());
      WS_STAMP(4);
      __syncthreads();                         // (B) the consumers have finished reading this item's image
      WS_STAMP(5);
      if (more) deposit();
      WS_STAMP(6);
      tile = ntile;
      chunk = nchunk;
    }
    return;
  }

  // ================================================================================================== consumer waves
  const int half = lq & 1, tp = lq >> 1;
  uint4 wv[WU];
  auto load_w = [&](int ch, int dz) {      // WHOLE: dz = 0 and all three planes (they are contiguous in the workspace)
    const uint4* src = wsplit + (long)(ch * 3 + dz) * 6 * Bf3::SLOTS * g.CoutP + n0;
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      const int it = tid + i * NC;
      wv[i] = make_uint4(0u, 0u, 0u, 0u);
      if (it < WITEMS && n0 + it % NB < g.CoutP) wv[i] = src[(long)(it / NB) * g.CoutP + it % NB];
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      const int it = tid + i * NC;
      if (it < WITEMS) Ws[it] = wv[i];
    }
  };
  const int wz = (4 * wid) / TC::TY, wy = (4 * wid) % TC::TY;
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
  const uint4* xa = Xs + half * NPOSP + lanepos;
  const uint4* wb = Ws + (half * Bf3::SLOTS + tp) * NB + lr;

  f32x4 acc[MB][NBT];
  uint4 pa1[MB], pa23[MB][2], pb[PIPE_B2 ? 2 : 1][3][NBT];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < NBT; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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

  if (tile < g.ntiles) {
    load_w(0, 0);
    if (WHOLE) store_w();
  }
  // the bias of the lane's output channels, loaded ONCE: read inside the epilogue it is a global load whose latency nothing hides for
  // the late wave of a SIMD (round 5, stamps of tools/probe/conv_cl16_probe.hip: its epilogue behind barrier B took 2.0 k of a 16 k-cycle
  // item — the load, not the four stores)
  float bvs[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int co = n0 + j * 16 + lr;
    bvs[j] = (bias && co < g.Cout) ? bias[co] : 0.f;
    ICL_PIN1(bvs[j]);
  }
  bool first_item = true;
  int item_no = -1;
  (void)item_no;
  Bf3RunStats<NBT> run;                  // InstanceNorm statistics of this workgroup's outputs (g.stats, conv_bf16x3.h)
  run.reset();
  int my_sample = -1;
  while (tile < g.ntiles) {
    int ntile = tile, nchunk = chunk + 1;
    if (nchunk == g.nchunks) { nchunk = 0; ntile = next_tile(tile); }
    ++item_no;
    WS_STAMP(0);
    if (WHOLE && g.nchunks > 1) {
      // between the barriers B of the last item and A of this one: this chunk's weight planes go to LDS and the NEXT chunk's are
      // requested — here, in front of barrier A, not behind it: behind A the loaders fill the vector-memory queue with their 288
      // gather instructions and a consumer's six loads wait behind them (stamps: its first tap stage took 9.4 k cycles instead of 2.6 k)
      if (!first_item) store_w();
      if (ntile < g.ntiles) load_w(nchunk, 0);
    }
    first_item = false;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      if (WHOLE) {
        if (dz == 0) {
          __syncthreads();                     // (A)
          WS_STAMP(1);
        }
      } else {
        if (dz > 0) __syncthreads();           // the previous plane's weights are no longer read
        store_w();
        __syncthreads();                       // dz = 0: (A)
        if (dz < 2) load_w(chunk, dz + 1);
        else if (ntile < g.ntiles) load_w(nchunk, 0);
      }
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
      WS_STAMP(2 + dz);
    }
    // ---- epilogue.  The two consumer waves of a SIMD multiply one after the other (the older wave, wid < 4, first): the early wave
    // stores in front of barrier B, beside its partner's multiply; the late wave stores behind B, beside the loaders' deposit of the
    // next tile (1.7 k cycles of LDS writes that nothing else could overlap) — its accumulators do not touch LDS.
    auto epilogue = [&]() __attribute__((always_inline)) {
      const int b = tile / tiles_per, bt = tile % tiles_per;
      const int x0 = (bt % g.ntx) * TC::TX, y0 = ((bt / g.ntx) % g.nty) * TC::TY, z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
      float* yb = y + (long)b * g.y_bstride;
#pragma unroll
      for (int j = 0; j < NBT; ++j) {
        const int co = n0 + j * 16 + lr;
        const float bv = bvs[j];
        float sv[16];
        bool sok[4] = {false, false, false, false};
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          int gz, gy, gx;
          if (FLAT) {
            const int p = 16 * (MB * wid + m) + 4 * lq;
            gz = z0 + p / (TC::TY * TC::TX); gy = y0 + (p / TC::TX) % TC::TY; gx = x0 + p % TC::TX;
          } else {
            gz = z0 + wz; gy = y0 + wy + m; gx = x0 + 4 * lq;
          }
          const float4 v = make_float4(acc[m][j][0] + bv, acc[m][j][1] + bv, acc[m][j][2] + bv, acc[m][j][3] + bv);
          sv[4 * m] = v.x; sv[4 * m + 1] = v.y; sv[4 * m + 2] = v.z; sv[4 * m + 3] = v.w;
          sok[m] = co < g.Cout && gz < g.D && gy < g.H && gx < g.W;
          if (sok[m]) *reinterpret_cast<float4*>(yb + (long)co * DHW + gz * HW + (long)gy * g.W + gx) = v;
          acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (g.stats) bf3_stats_add(run, j, sv, sok, 4 * MB);
      }
      my_sample = b;
    };
    const bool last_chunk = chunk == g.nchunks - 1;
    if (last_chunk && wid < 4) epilogue();
    WS_STAMP(5);
    __syncthreads();                           // (B)
    WS_STAMP(6);
    if (last_chunk && wid >= 4) epilogue();
    tile = ntile;
    chunk = nchunk;
  }
  // (the loader waves have returned: the barrier inside counts the eight consumer waves)
  if (g.stats)
    bf3_stats_flush<NBT, 8>(run, reinterpret_cast<float*>(Ws + WPL * TC::ws_u4(NB)), g.stats, my_sample, g.nbatch, g.Cout, n0, g.wgs,
                            (int)blockIdx.x, wid, lane, tid);
}

}  // namespace icl
