// conv_bf16x3_zr.h — the loader-wave split-product 3x3x3 convolution (conv_bf16x3_ws.h) for ONE 16-channel chunk and one cout block,
// walking z-columns with the halo tile kept in a RING of its six z-planes (round 5).
//
// conv3d_bf16x3_fwd_ws_kernel stages the 6 x 10 x 18 halo positions of every 4 x 8 x 16 tile; consecutive tiles of a workgroup are far
// apart.  Stamps of round 5 (profiles/r5_cl16_stage_a.md) price an item of that kernel at 12.2 k cycles of multiply + 2.2 k during which
// the consumers wait for the loaders' split to finish + 1.6 k for the loaders' deposit of 104 KB into LDS.  With Cin = 16 there is one
// chunk per tile, so a workgroup can take consecutive tiles of a z-COLUMN: the next tile's planes pz = 0, 1 are this tile's planes
// pz = 4, 5, already split and already in LDS.  Here plane pz of the tile at column index tz lives in plane slot (4 tz + pz) mod 6, the
// loaders fetch, split and deposit only the FOUR planes a tile adds (24 of 36 staging rounds per wave: the split fits inside the multiply
// window, the deposit is a third shorter), and the consumers address a tap's plane through one of three per-item base offsets instead of
// an immediate.  A column's first tile (and a run's first tile) is staged whole.  Same LDS images, weight planes, MFMA order and epilogue
// as conv3d_bf16x3_fwd_ws_kernel<1>: bit-identical outputs and statistics summaries of equal content (the tile order differs, so the
// summaries sit in other slots).
// Tiles are numbered z-fastest; a workgroup owns a contiguous run of g.tiles_per_wg tiles INSIDE one sample (the statistics need that),
// and when the grid is a multiple of 8 the runs of one XCD are adjacent columns.
// Reference op: nn.Conv3d(k=3, pad=1) inside UnetConv3 (/root/reference/code/networks/utils.py:104,107) and its input gradient.
#pragma once

namespace icl {

__global__ __launch_bounds__(768) void conv3d_bf16x3_fwd_zr_kernel(const float* __restrict__ x, const uint4* __restrict__ wsplit,
                                                                   const float* __restrict__ bias, float* __restrict__ y, Bf3Geom g) {
  typedef Bf3T<8> TC;
  constexpr int MB = 4, NB = 16, PX = TC::PX, PY = TC::PY, NPOSP = TC::NPOSP, PLANE = TC::PY * TC::PX;      // 180 positions per z-plane
  constexpr int NC = 512, NL = 256;                     // consumer / loader threads
  constexpr int WITEMS = 3 * 6 * Bf3::SLOTS * NB, WU = (WITEMS + NC - 1) / NC;
  // loader staging rounds: item = (channel octet, position); rounds [0, RN) the four NEW planes pz = 2..5, rounds [RN, RN + RX) the two
  // inherited ones (staged only for a column's / run's first tile)
  constexpr int NEW_ITEMS = 2 * 4 * PLANE, OLD_ITEMS = 2 * 2 * PLANE;
  constexpr int RN = (NEW_ITEMS + NL - 1) / NL, RX = (OLD_ITEMS + NL - 1) / NL, ROUNDS = RN + RX;
  ICL_DYN_LDS(uint4, lds);
  uint4* Xs = lds;
  uint4* Ws = lds + TC::XS_U4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  int wid = tid >> 6;
  ICL_WAVE_UNIFORM(wid);
  const int n0 = blockIdx.y * NB;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const int tiles_per = g.ntz * g.nty * g.ntx;          // per sample

  // zero the pad positions once (read by the zero slot: garbage * 0 must not be NaN)
  for (int i = tid; i < 6 * (NPOSP - TC::NPOS); i += NC + NL)
    Xs[(i / (NPOSP - TC::NPOS)) * NPOSP + TC::NPOS + (i % (NPOSP - TC::NPOS))] = make_uint4(0u, 0u, 0u, 0u);

  // workgroup -> (sample, run of tiles inside the sample); the runs of one XCD are adjacent
  const int wgl = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int runs_per = (tiles_per + g.tiles_per_wg - 1) / g.tiles_per_wg;
  const int my_b = wgl / runs_per, my_run = wgl % runs_per;
  int t_begin = my_b * tiles_per + my_run * g.tiles_per_wg;
  int t_end = t_begin + g.tiles_per_wg < (my_b + 1) * tiles_per ? t_begin + g.tiles_per_wg : (my_b + 1) * tiles_per;
  if (my_b >= g.nbatch) t_begin = t_end = 0;            // (padding workgroups of a grid rounded up to a multiple of 8)
  // tile -> column index along z and origin
  auto decode = [&](int t, int& tz, int& oz, int& oy, int& ox) {
    const int bt = t - my_b * tiles_per, col = bt / g.ntz;
    tz = bt - col * g.ntz;
    oz = tz * TC::TZ; ox = (col % g.ntx) * TC::TX; oy = (col / g.ntx) * TC::TY;
  };

  if (wid >= 8) {
    // ================================================================================================ loader waves
    const int lt = tid - NC;
    int s_zd[ROUNDS], s_rel[ROUNDS];                     // (pz, py, px, LDS offset without the plane slot) packed; source offset relative to the tile origin
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const bool fresh_round = r >= RN;
      const int it = lt + (fresh_round ? r - RN : r) * NL, per_o = (fresh_round ? 2 : 4) * PLANE;
      const int o = it / per_o, rem = it % per_o;
      const int pz = (fresh_round ? 0 : 2) + rem / PLANE, pos = rem % PLANE, py = pos / PX, px = pos % PX;
      const bool live = it < 2 * per_o;
      s_zd[r] = live ? (pz << 27) | (py << 22) | (px << 16) | (o * NPOSP + pos) : -1;
      s_rel[r] = o * 8 * (int)DHW + (pz - 1) * (int)HW + (py - 1) * g.W + (px - 1);
    }
    uint4 pl[ROUNDS][3];                                 // the three packed planes of every staging item of the tile in flight
    icl_rsrc_t xr = icl_make_rsrc(x + (long)my_b * g.x_bstride, (unsigned)(16 * DHW * 4));
    int toff = 0, oz = 0, oy = 0, ox = 0, ntz_ = 0;
    auto issue = [&](auto R0, auto R1) __attribute__((always_inline)) {
#pragma unroll
      for (int r = decltype(R0)::value; r < decltype(R1)::value; ++r) {
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
      for (int r = decltype(R0)::value; r < decltype(R1)::value; ++r) {
        const float v[8] = {__uint_as_float(pl[r][0].x), __uint_as_float(pl[r][0].y), __uint_as_float(pl[r][0].z), __uint_as_float(pl[r][0].w),
                            __uint_as_float(pl[r][1].x), __uint_as_float(pl[r][1].y), __uint_as_float(pl[r][1].z), __uint_as_float(pl[r][1].w)};
        bf3_split8(v, pl[r][0], pl[r][1], pl[r][2]);
      }
    };
    // plane pz of the tile at column index tz sits in plane slot (4 tz + pz) mod 6
    auto deposit = [&](auto R0, auto R1) __attribute__((always_inline)) {
      const int base = (4 * ntz_) % 6;
#pragma unroll
      for (int r = decltype(R0)::value; r < decltype(R1)::value; ++r) {
        if (s_zd[r] < 0) continue;
        int ps = base + ((s_zd[r] >> 27) & 15);
        ps = ps >= 6 ? ps - 6 : ps;
        uint4* d = Xs + (s_zd[r] & 0xffff) + ps * PLANE;
        d[0] = pl[r][0];
        d[2 * NPOSP] = pl[r][1];
        d[4 * NPOSP] = pl[r][2];
      }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, RN / 3> I1;
    typedef std::integral_constant<int, RN> I2;
    typedef std::integral_constant<int, ROUNDS> I3;
    auto origin = [&](int t) {
      decode(t, ntz_, oz, oy, ox);
      toff = oz * (int)HW + oy * g.W + ox;
    };
    int tile = t_begin;
    if (tile < t_end) {                                  // a run's first tile: all six planes
      origin(tile);
      issue(I0(), I3());
      split_range(I0(), I3());
      deposit(I0(), I3());
    }
    while (tile < t_end) {
      const int ntile = tile + 1;
      const bool more = ntile < t_end;
      bool fresh = false;
      __syncthreads();                         // (A) the LDS image of this item (and the consumers' weights) is complete
      if (more && !(g.flags & 8)) {            // (flags bit 3: timing ablation — no staging after the first tile)
        origin(ntile);
        fresh = ntz_ == 0 && !(g.flags & 4);   // the next tile starts a column: nothing to inherit (flags bit 2: timing ablation, wrong results)
        issue(I0(), I2());
        if (fresh) issue(I2(), I3());
        split_range(I0(), I1());
        split_range(I1(), I2());
        if (fresh) split_range(I2(), I3());
      }
      __syncthreads();                         // (B) the consumers have finished reading this item's image
      if (more && !(g.flags & 8)) {
        deposit(I0(), I2());
        if (fresh) deposit(I2(), I3());
      }
      tile = ntile;
    }
    return;
  }

  // ================================================================================================== consumer waves
  const int half = lq & 1, tp = lq >> 1;
  uint4 wv[WU];
  if (t_begin < t_end) {                                 // all three weight planes of the one chunk: loaded once per workgroup
    const uint4* src = wsplit + n0;
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      const int it = tid + i * NC;
      wv[i] = make_uint4(0u, 0u, 0u, 0u);
      if (it < WITEMS && n0 + it % NB < g.CoutP) wv[i] = src[(long)(it / NB) * g.CoutP + it % NB];
    }
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      const int it = tid + i * NC;
      if (it < WITEMS) Ws[it] = wv[i];
    }
  }
  const int wz = (4 * wid) / TC::TY, wy = (4 * wid) % TC::TY;
  const uint4* wb = Ws + (half * Bf3::SLOTS + tp) * NB + lr;

  f32x4 acc[MB];
  uint4 pa1[MB], pa23[MB][2], pb[2][3];
#pragma unroll
  for (int m = 0; m < MB; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment pointers of the 14 tap pairs: per ITEM values here (the plane slot of a tap's dz moves with the ring) — computed in front of
  // barrier A, where the consumers wait for the loaders anyway: inside the multiply loop they would be vector instructions between the
  // MFMAs (measured: 128 instead of ~105 us for the multiply alone)
  int fo[3][5];
  auto frag_ptr = [&](int sdz, int spair) { return Xs + fo[sdz][spair]; };
  auto load_b = [&](int bi, int sdz, int spair, int s0 = 0, int s1 = 3) {
#pragma unroll
    for (int s = s0; s < s1; ++s) pb[bi][s] = wb[((sdz * 6) * Bf3::SLOTS + s * 2 * Bf3::SLOTS + spair * 2) * NB];
  };
  auto load_x1 = [&](int sdz, int spair) {
    const uint4* xp = frag_ptr(sdz, spair);
#pragma unroll
    for (int m = 0; m < MB; ++m) pa1[m] = xp[m * PX];
  };
  auto load_x23 = [&](int sdz, int spair) {
    const uint4* xp = frag_ptr(sdz, spair);
#pragma unroll
    for (int m = 0; m < MB; ++m) pa23[m][1] = xp[4 * NPOSP + m * PX];      // a3 first: its products lead the Y half
#pragma unroll
    for (int m = 0; m < MB; ++m) pa23[m][0] = xp[2 * NPOSP + m * PX];
  };

  float bv = (bias && n0 + lr < g.Cout) ? bias[n0 + lr] : 0.f;
  ICL_PIN1(bv);
  Bf3RunStats<1> run;
  run.reset();
  for (int tile = t_begin; tile < t_end; ++tile) {
    int tz, z0, y0, x0;
    decode(tile, tz, z0, y0, x0);
    {
      int zoff[3];
      const int base = (4 * tz) % 6 + wz;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const int ps = base + dz;
        zoff[dz] = (ps >= 6 ? ps - 6 : ps) * PLANE;
      }
      const int lane0 = half * NPOSP + wy * PX + lr;
#pragma unroll
      for (int sdz = 0; sdz < 3; ++sdz)
#pragma unroll
        for (int spair = 0; spair < 5; ++spair) {
          const int tA = 10 * sdz + 2 * spair < 27 ? 10 * sdz + 2 * spair : 26, tB = tA + 1 < 27 ? tA + 1 : 26;
          const int offA = ((tA / 3) % 3) * PX + tA % 3, offB = ((tB / 3) % 3) * PX + tB % 3;
          fo[sdz][spair] = lane0 + (tp ? zoff[tB / 9] + offB : zoff[tA / 9] + offA);
          ICL_OPAQUE_INT(fo[sdz][spair]);          // (a value, not an expression the scheduler may re-materialise between the MFMAs)
        }
    }
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      if (dz == 0) __syncthreads();                      // (A)
      const int np = dz < 2 ? 5 : 4;
      if (dz == 0) {
        load_b(0, 0, 0);
        load_x1(0, 0);
      }
#pragma unroll
      for (int pair = 0; pair < np; ++pair) {
        const int cur = (5 * dz + pair) & 1;
        load_x23(dz, pair);
#pragma unroll
        for (int sb = 2; sb >= 0; --sb)
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[m] = icl_mfma_16x16x32_bf16(pa1[m], pb[cur][sb], acc[m]);
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
          ICL_SCHED_GROUP(0x008, 1);
          ICL_SCHED_GROUP(0x100, 1);
        }
        ICL_SCHED_GROUP(0x008, MB);
        ICL_SCHED_BARRIER();
        const bool more = pair + 1 < np || dz < 2;
        const int ndz = pair + 1 < np ? dz : dz + 1, npair = pair + 1 < np ? pair + 1 : 0;
        if (more) {
          load_b(cur ^ 1, ndz, npair, 2, 3);
          load_x1(ndz, npair);
          load_b(cur ^ 1, ndz, npair, 0, 2);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          constexpr int sa[3] = {1, 0, 0}, sbb[3] = {0, 1, 0};      // a3 b1, a2 b2, a2 b1
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[m] = icl_mfma_16x16x32_bf16(pa23[m][sa[t]], pb[cur][sbb[t]], acc[m]);
        }
        if (more) {
          constexpr int R = MB + 3;                            // reads of this half-step
          constexpr int NM = 3 * MB;                           // its MFMAs
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
    // ---- epilogue (conv_bf16x3_ws.h): the early wave of a SIMD stores in front of barrier B, the late one behind it
    auto epilogue = [&]() __attribute__((always_inline)) {
      float* yb = y + (long)my_b * g.y_bstride;
      const int co = n0 + lr;
      float sv[16];
      bool sok[4] = {false, false, false, false};
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int gz = z0 + wz, gy = y0 + wy + m, gx = x0 + 4 * lq;
        const float4 v = make_float4(acc[m][0] + bv, acc[m][1] + bv, acc[m][2] + bv, acc[m][3] + bv);
        sv[4 * m] = v.x; sv[4 * m + 1] = v.y; sv[4 * m + 2] = v.z; sv[4 * m + 3] = v.w;
        sok[m] = co < g.Cout && gz < g.D && gy < g.H && gx < g.W;
        if (sok[m]) *reinterpret_cast<float4*>(yb + (long)co * DHW + gz * HW + (long)gy * g.W + gx) = v;
        acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (g.stats) bf3_stats_add(run, 0, sv, sok, 4 * MB);
    };
    if (wid < 4) epilogue();
    __syncthreads();                           // (B)
    if (wid >= 4) epilogue();
  }
  // (the loader waves have returned: the barrier inside counts the eight consumer waves)
  if (g.stats)
    bf3_stats_flush<1, 8>(run, reinterpret_cast<float*>(Ws + 3 * TC::ws_u4(NB)), g.stats, my_b < g.nbatch ? my_b : -1, g.nbatch, g.Cout, n0, g.wgs,
                          wgl, wid, lane, tid);
}

}  // namespace icl
