// conv_bf16x3_wsr.h — PROBE (round 4, measured and not adopted: profiles/r4_wsr_probe.txt: 0.89-0.99x of conv3d_bf16x3_fwd_kernel<2|3, 8, 60>).
// Loader waves for two / three cout blocks: the consumers run the one-block multiply loop once per cout block over the same LDS image
// (so that they stay under the 168 registers three waves per SIMD allow), weight planes through a three-slot LDS-DMA ring issued two
// stages ahead with counted s_waitcnt and raw s_barrier.  Bit-identical to the shipped kernel and race-free in ten repeats — and slower:
// the single-role kernel reads every X fragment from LDS once for all its cout blocks (48->48 @96^3: 240 TFLOP/s = 0.58 of its roofline
// in isolation), this one once per block, with a barrier per (cout block, dz) stage on top.
#pragma once

namespace icl {

// ------------------------------------------------------------------------------------------------ more than one cout block
// conv3d_bf16x3_fwd_ws_kernel needs consumers of <= 168 registers (three waves per SIMD); with two / three cout blocks per wave the
// multiply loop of conv_bf16x3.h takes 220-250.  Here a workgroup still owns NCBLK cout blocks of a tile — ONE halo fetch per item,
// by the loader waves — but the consumers run the one-block multiply loop NCBLK times over the same LDS image, one cout block after
// the other (NCBLK x 16 accumulator registers; the X fragments are read from LDS once per block: the one-block kernel's LDS load).
// The weight planes (one (cout block, dz) plane = 15 KB) travel through a ring of THREE LDS slots by LDS-DMA, issued by the consumer
// waves two stages ahead (a stage = one dz plane of one cout block, ~3.6 k cycles; a load issued while the loaders fetch comes back
// after ~3 us): no registers, no ds_write; a counted s_waitcnt leaves the later plane in flight and the stage barrier is a raw
// s_barrier (ICL_BARRIER_KEEP_VMEM: __syncthreads would drain the DMA).  The loaders deal their 72 gathers and the split over the
// stages so that they reach every stage barrier on time.  Same summation order per output as conv3d_bf16x3_fwd_kernel: bit-identical.
template <int NCBLK, bool FLAT = false>
__global__ __launch_bounds__(768) void conv3d_bf16x3_fwd_wsr_kernel(const float* __restrict__ x, const uint4* __restrict__ wsplit,
                                                                    const float* __restrict__ bias, float* __restrict__ y, Bf3Geom g) {
  typedef typename std::conditional<FLAT, Bf3F24, Bf3T<8>>::type TC;
  constexpr int MB = FLAT ? 3 : 4;
  constexpr int S = 3 * NCBLK;                          // stages per work item
  constexpr int PX = TC::PX, PY = TC::PY, NPOSP = TC::NPOSP;
  constexpr int NC = 512, NL = 256;
  constexpr int PLANE = 6 * Bf3::SLOTS * 16;            // uint4 per (cout block, dz) weight plane: 960 = 15 DMA blocks of 64
  constexpr int ITEMS = 2 * TC::NPOS, ROUNDS = (ITEMS + NL - 1) / NL;
  ICL_DYN_LDS(uint4, lds);
  uint4* Xs = lds;
  uint4* Wr = lds + TC::XS_U4;                          // ring of three weight planes
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  int wid = tid >> 6;
  ICL_WAVE_UNIFORM(wid);
  const int n0 = blockIdx.y * 16 * NCBLK;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const int tiles_per = g.ntz * g.nty * g.ntx;

  for (int i = tid; i < 6 * (NPOSP - TC::NPOS); i += NC + NL)
    Xs[(i / (NPOSP - TC::NPOS)) * NPOSP + TC::NPOS + (i % (NPOSP - TC::NPOS))] = make_uint4(0u, 0u, 0u, 0u);

  const int per_xcd = (g.ntiles + 7) / 8, xcd = blockIdx.x & 7, wgs_per_xcd = gridDim.x >> 3;
  const int xcd_end = (xcd + 1) * per_xcd < g.ntiles ? (xcd + 1) * per_xcd : g.ntiles;
  auto next_tile = [&](int t) { return t + wgs_per_xcd < xcd_end ? t + wgs_per_xcd : g.ntiles; };
  int tile = xcd * per_xcd + (blockIdx.x >> 3), chunk = 0;
  if (tile >= xcd_end) tile = g.ntiles;

  if (wid >= 8) {
    // ================================================================================================ loader waves
    const int lt = tid - NC;
    ICL_SETPRIO(0);
    int s_zd[ROUNDS], s_rel[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int it = lt + r * NL;
      const int o = it / TC::NPOS, pos = it % TC::NPOS;
      const int px = pos % PX, row = pos / PX, py = row % PY, pz = row / PY;
      s_zd[r] = it < ITEMS ? (pz << 27) | (py << 22) | (px << 16) | (o * NPOSP + pos) : -1;
      s_rel[r] = o * 8 * (int)DHW + (pz - 1) * (int)HW + (py - 1) * g.W + (px - 1);
    }
    uint4 pl[ROUNDS][3];
    icl_rsrc_t xr = icl_make_rsrc(x, 0u);
    int toff = 0, oz = 0, oy = 0, ox = 0;
    auto origin = [&](int t, int ch) {
      const int b = t / tiles_per, bt = t % tiles_per;
      ox = (bt % g.ntx) * TC::TX; oy = ((bt / g.ntx) % g.nty) * TC::TY; oz = (bt / (g.ntx * g.nty)) * TC::TZ;
      xr = icl_make_rsrc(x + (long)b * g.x_bstride + (long)ch * 16 * DHW, (unsigned)(16 * DHW * 4));
      toff = oz * (int)HW + oy * g.W + ox;
    };
    auto issue = [&](auto R0, auto R1) __attribute__((always_inline)) {
#pragma unroll
      for (int r = decltype(R0)::value; r < decltype(R1)::value; ++r) {
        if (r >= ROUNDS) continue;
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
    auto split = [&](auto R0, auto R1) __attribute__((always_inline)) {
#pragma unroll
      for (int r = decltype(R0)::value; r < decltype(R1)::value; ++r) {
        if (r >= ROUNDS) continue;
        const float v[8] = {__uint_as_float(pl[r][0].x), __uint_as_float(pl[r][0].y), __uint_as_float(pl[r][0].z), __uint_as_float(pl[r][0].w),
                            __uint_as_float(pl[r][1].x), __uint_as_float(pl[r][1].y), __uint_as_float(pl[r][1].z), __uint_as_float(pl[r][1].w)};
        bf3_split8(v, pl[r][0], pl[r][1], pl[r][2]);
      }
    };
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
    // rounds issued in stage s: the first S - 2 stages share them out (so that every load has two stages to arrive before its split,
    // and a loader reaches each stage barrier after a bounded piece of work); split in stage s: the rounds issued in stage s - 2
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, ROUNDS> IR;
    if (tile < g.ntiles) {
      origin(tile, 0);
      issue(I0(), IR());
      split(I0(), IR());
      deposit();
    }
    while (tile < g.ntiles) {
      int ntile = tile, nchunk = chunk + 1;
      if (nchunk == g.nchunks) { nchunk = 0; ntile = next_tile(tile); }
      const bool more = ntile < g.ntiles;
      if (more) origin(ntile, nchunk);
#define WSR_LOADER_STAGE(s)                                                                                      \
      ICL_BARRIER_KEEP_VMEM();                                                                                      \
      if (more) {                                                                                                   \
        constexpr int a0 = (s) < S - 2 ? (ROUNDS * (s) + S - 3) / (S - 2) : ROUNDS;                                 \
        constexpr int a1 = (s) + 1 < S - 2 ? (ROUNDS * ((s) + 1) + S - 3) / (S - 2) : ROUNDS;                        \
        constexpr int b0 = (s) >= 2 ? ((s) - 2 < S - 2 ? (ROUNDS * ((s) - 2) + S - 3) / (S - 2) : ROUNDS) : 0;       \
        constexpr int b1 = (s) >= 2 ? ((s) - 1 < S - 2 ? (ROUNDS * ((s) - 1) + S - 3) / (S - 2) : ROUNDS) : 0;       \
        issue(std::integral_constant<int, a0>(), std::integral_constant<int, a1>());                                \
        split(std::integral_constant<int, b0>(), std::integral_constant<int, b1>());                                \
      }
      WSR_LOADER_STAGE(0) WSR_LOADER_STAGE(1) WSR_LOADER_STAGE(2) WSR_LOADER_STAGE(3) WSR_LOADER_STAGE(4) WSR_LOADER_STAGE(5)
      if constexpr (NCBLK == 3) { WSR_LOADER_STAGE(6) WSR_LOADER_STAGE(7) WSR_LOADER_STAGE(8) }
#undef WSR_LOADER_STAGE
      ICL_BARRIER_KEEP_VMEM();                 // (B) the consumers have finished reading this item's image
      if (more) deposit();
      tile = ntile;
      chunk = nchunk;
    }
    return;
  }

  // ================================================================================================== consumer waves
  const int half = lq & 1, tp = lq >> 1;
  const int nw_dma = wid < 7 ? 2 : 1;                   // this wave's DMA instructions per weight plane (blocks wid, wid + 8 of 15)
  (void)nw_dma;
  // one weight plane into ring slot `slot`: rows (q, tap slot) x 16 couts of cout block cb, dz plane dz of channel chunk ch
  auto issue_plane = [&](int ch, int cb, int dz, int slot) {
    // (a cout block beyond the padded extent — a ragged last workgroup column — fetches block 0's plane: its products are never stored,
    // and the counted waits below rely on every stage issuing the same number of DMA instructions)
    const int cbe = n0 + 16 * cb < g.CoutP ? cb : 0;
    const uint4* src = wsplit + ((long)(ch * 3 + dz) * 6 * Bf3::SLOTS) * g.CoutP + n0 + 16 * cbe;
    const icl_rsrc_t wr = icl_make_rsrc(src, (unsigned)((6 * Bf3::SLOTS * g.CoutP - n0 - 16 * cbe) * 16));
    const unsigned vo = (unsigned)((lane >> 4) * g.CoutP + (lane & 15)) * 16u;
    icl_buffer_load_lds_b128(wr, Wr + slot * PLANE + wid * 64, vo, (unsigned)(wid * 4 * g.CoutP) * 16u);
    if (wid < 7) icl_buffer_load_lds_b128(wr, Wr + slot * PLANE + (wid + 8) * 64, vo, (unsigned)((wid + 8) * 4 * g.CoutP) * 16u);
  };
  // stage index of an item -> (cout block, dz); the plane of stage s of the item AFTER (tile, chunk)
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

  float bv[NCBLK];                                      // fetched before the first DMA is in flight (see conv_planes.h)
#pragma unroll
  for (int cb = 0; cb < NCBLK; ++cb) {
    const int co = n0 + cb * 16 + lr;
    bv[cb] = (bias && co < g.Cout) ? bias[co] : 0.f;
    ICL_PIN1(bv[cb]);
  }
  f32x4 acc[NCBLK][MB];
  uint4 pa1[MB], pa23[MB][2], pb[2][3];
#pragma unroll
  for (int cb = 0; cb < NCBLK; ++cb)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[cb][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  Bf3RunStats<NCBLK> run;
  run.reset();
  int my_sample = -1;
  if (tile < g.ntiles) {
    issue_plane(0, 0, 0, 0);
    issue_plane(0, 0, 1, 1);
  }
  bool prev_epilogue = false;
  while (tile < g.ntiles) {
    int ntile = tile, nchunk = chunk + 1;
    if (nchunk == g.nchunks) { nchunk = 0; ntile = next_tile(tile); }
    const bool more = ntile < g.ntiles;
    auto frag_ptr = [&](int sdz, int spair) {
      const int tA = 10 * sdz + 2 * spair, tB = tA + 1 < 27 ? tA + 1 : 26;
      const int offA = (tA / 9) * PY * PX + ((tA / 3) % 3) * PX + tA % 3, offB = (tB / 9) * PY * PX + ((tB / 3) % 3) * PX + tB % 3;
      return xa + (tp ? offB : offA);
    };
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int cb = s / 3, dz = s % 3;
      // plane s has landed: younger operations of this wave = the plane of stage s + 1 (nw_dma instructions) — and, at stage 0 behind
      // an epilogue, that item's output stores, whose count the compiler may have changed by branching around masked ones: wait for all
      // (and nothing younger exists at the last stage of the last item)
      if ((s == 0 && prev_epilogue) || (s == S - 1 && !more)) ICL_WAIT_VMEM();
      else if (wid < 7) ICL_WAIT_VMCNT(2);
      else ICL_WAIT_VMCNT(1);
      ICL_BARRIER_KEEP_VMEM();                 // stage 0: (A) the loaders' image of this item is complete
      // the plane of stage s + 2 into the slot that stage s - 1 has just released
      if (s + 2 < S) issue_plane(chunk, (s + 2) / 3, (s + 2) % 3, (s + 2) % 3);
      else if (more) issue_plane(nchunk, 0, s + 2 - S, (s + 2) % 3);
      const uint4* wb = Wr + (s % 3) * PLANE + (half * Bf3::SLOTS + tp) * 16 + lr;
      auto load_b = [&](int bi, int spair, int s0 = 0, int s1 = 3) {
#pragma unroll
        for (int sp = s0; sp < s1; ++sp) pb[bi][sp] = wb[(sp * 2 * Bf3::SLOTS + spair * 2) * 16];
      };
      auto load_x1 = [&](int spair) {
        const uint4* xp = frag_ptr(dz, spair);
#pragma unroll
        for (int m = 0; m < MB; ++m) pa1[m] = xp[moff[m]];
      };
      auto load_x23 = [&](int spair) {
        const uint4* xp = frag_ptr(dz, spair);
#pragma unroll
        for (int m = 0; m < MB; ++m) pa23[m][1] = xp[4 * NPOSP + moff[m]];
#pragma unroll
        for (int m = 0; m < MB; ++m) pa23[m][0] = xp[2 * NPOSP + moff[m]];
      };
      const int np = dz < 2 ? 5 : 4;
      load_b(0, 0);
      load_x1(0);
#pragma unroll
      for (int pair = 0; pair < np; ++pair) {
        const int cur = pair & 1;
        load_x23(pair);
#pragma unroll
        for (int sb = 2; sb >= 0; --sb)
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[cb][m] = icl_mfma_16x16x32_bf16(pa1[m], pb[cur][sb], acc[cb][m]);
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
          ICL_SCHED_GROUP(0x008, 1);
          ICL_SCHED_GROUP(0x100, 1);
        }
        ICL_SCHED_GROUP(0x008, MB);
        ICL_SCHED_BARRIER();
        const bool morep = pair + 1 < np;
        if (morep) {
          load_b(cur ^ 1, pair + 1, 2, 3);
          load_x1(pair + 1);
          load_b(cur ^ 1, pair + 1, 0, 2);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          constexpr int sa[3] = {1, 0, 0}, sbb[3] = {0, 1, 0};      // a3 b1, a2 b2, a2 b1
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[cb][m] = icl_mfma_16x16x32_bf16(pa23[m][sa[t]], pb[cur][sbb[t]], acc[cb][m]);
        }
        if (morep) {
          constexpr int R = MB + 3, NM = 3 * MB;
#pragma unroll
          for (int i = 0; i < (R < NM ? R : NM); ++i) {
            ICL_SCHED_GROUP(0x008, 1);
            ICL_SCHED_GROUP(0x100, 1);
          }
          if (NM > R) ICL_SCHED_GROUP(0x008, NM - R);
        }
        ICL_SCHED_BARRIER();
      }
    }
    // ---- epilogue (early wave of a SIMD in front of barrier B, late wave behind it: conv3d_bf16x3_fwd_ws_kernel)
    auto epilogue = [&]() __attribute__((always_inline)) {
      const int b = tile / tiles_per, bt = tile % tiles_per;
      const int x0 = (bt % g.ntx) * TC::TX, y0 = ((bt / g.ntx) % g.nty) * TC::TY, z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
      float* yb = y + (long)b * g.y_bstride;
      const bool nt = (g.flags & 2) != 0;
#pragma unroll
      for (int cb = 0; cb < NCBLK; ++cb) {
        const int co = n0 + cb * 16 + lr;
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
          const float4 v = make_float4(acc[cb][m][0] + bv[cb], acc[cb][m][1] + bv[cb], acc[cb][m][2] + bv[cb], acc[cb][m][3] + bv[cb]);
          sv[4 * m] = v.x; sv[4 * m + 1] = v.y; sv[4 * m + 2] = v.z; sv[4 * m + 3] = v.w;
          sok[m] = co < g.Cout && gz < g.D && gy < g.H && gx < g.W;
          if (sok[m]) {
            float* dst = yb + (long)co * DHW + gz * HW + (long)gy * g.W + gx;
            if (nt) icl_nt_store4(dst, v); else *reinterpret_cast<float4*>(dst) = v;
          }
          acc[cb][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (g.stats) bf3_stats_add(run, cb, sv, sok, 4 * MB);
      }
      my_sample = b;
    };
    const bool last_chunk = chunk == g.nchunks - 1;
    if (last_chunk && wid < 4) epilogue();
    ICL_BARRIER_KEEP_VMEM();                   // (B)
    if (last_chunk && wid >= 4) epilogue();
    prev_epilogue = last_chunk;
    tile = ntile;
    chunk = nchunk;
  }
  if (g.stats)
    bf3_stats_flush<NCBLK, 8>(run, reinterpret_cast<float*>(Wr + 3 * PLANE), g.stats, my_sample, g.nbatch, g.Cout, n0, (int)gridDim.x,
                              (int)blockIdx.x, wid, lane, tid);
}

}  // namespace icl
