// conv_wgrad_tr_ws.h — the transposing-read weight gradient (csrc/kernels/conv_wgrad_tr.h) with dedicated producer waves (round 4 probe).
//
// conv3d_wgrad_tr_kernel makes every wave both stage and multiply (S -> M or M -> S, the two waves of a SIMD in opposite phases): a
// phase is ~6.0 k cycles for 2.6 k cycles of matrix work per SIMD (stamps of round 3: split + store 2.1 k, issuing the loads 1.5 k,
// multiply 2.3-2.6 k — one serial chain per wave).  Here the roles are separate waves:
//   waves 0..7   consumers: wave (kg, th) multiplies k-step kg / tap half th of the tile in buffer t & 1 — nothing else;
//   waves 8..11  producers (issue priority 0: they fill the issue gaps of the consumers): split + store tile t + 1 into the other
//                buffer from registers whose loads were issued TWO phases earlier, then issue the loads of tile t + 3.
// One barrier per tile (all twelve waves).  LDS images, MFMA order, slab layout: those of conv3d_wgrad_tr_kernel — bit-identical results.
#pragma once

namespace icl {

#if !defined(WSW_PRODUCERS)
#define WSW_PRODUCERS 4
#endif
#if !defined(WSW_PRIO)
#define WSW_PRIO 0
#endif
#if defined(WGTR_WS_STAMPS)
// consumer waves 0 and 4 (one SIMD) and producer wave 8 of workgroup 0 record s_memtime at the boundaries of phases 4..7
__device__ long long g_wgtr_ws_stamps[3 * 4 * 8];
#define WSW_STAMP(k)                                                                                        \
  do {                                                                                                      \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (wid & 3) == 0 && lane == 0 && phase_no >= 4 && phase_no < 8) \
      g_wgtr_ws_stamps[((wid >> 2) * 4 + phase_no - 4) * 8 + (k)] = clock64();                               \
  } while (0)
#else
#define WSW_STAMP(k) ((void)0)
#endif

template <int NCB>
__global__ __launch_bounds__(512 + 64 * WSW_PRODUCERS) void conv3d_wgrad_tr_ws_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                              float* __restrict__ gwp, Bf3WGeom g) {
  typedef WgTrT<NCB> C;
  ICL_DYN_LDS(uint4, lds);
  const int tid = threadIdx.x, lane = tid & 63;
  int wid = tid >> 6;
  ICL_WAVE_UNIFORM(wid);
  const bool producer = wid >= 8;
  const int ptid = tid - 512;                            // producer thread index (256 of them)
  constexpr int PNT = 64 * WSW_PRODUCERS, PROUNDS = (C::ITEMS + PNT - 1) / PNT;
  const int kg = wid & 3, th = (wid >> 2) & 1;           // consumers: k-step of the tile, tap half
  const int lg = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int ncb = (g.CinP + 15) / 16;
  const int co0 = (blockIdx.y / ncb) * 16 * NCB, c0 = (blockIdx.y % ncb) * 16;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const int tiles_per = g.ntz * g.nty * g.ntx;

  // ---- staging tables (tile-invariant).  Item it = tid + r * NT: kind (halo x / tile dY) is uniform per (wave, round).
  int s_zyx[PROUNDS], s_dst[PROUNDS], s_rel[PROUNDS];
  bool s_isx[PROUNDS];
#pragma unroll
  for (int r = 0; r < PROUNDS; ++r) {
    s_zyx[r] = -1; s_dst[r] = 0; s_rel[r] = 0; s_isx[r] = false;
    if (!producer) continue;
    const int it = ptid + r * PNT;
    int isx = ((wid - 8) * 64 + r * PNT) < C::XPAD;
    ICL_WAVE_UNIFORM(isx);
    s_isx[r] = isx != 0;
    if (isx) {
      const int o = it / C::NPOS, pos = it % C::NPOS;
      const int px = pos % C::PX, row = pos / C::PX, py = row % C::PY, pz = row / C::PY;
      const bool live = it < C::XITEMS;
      s_zyx[r] = live ? (pz << 16) | (py << 8) | px : -1;
      s_dst[r] = live ? o * 3 * C::NPOSP + pos : C::NPOS;                  // idle lanes: a pad slot (never read)
      s_rel[r] = o * 8 * (int)DHW + pz * (int)HW + py * g.W + px;          // + tile origin - (1, 1, 1)
    } else {
      const int ig = it - C::XPAD, ob = ig / C::TPOS, pos = ig % C::TPOS;
      const int tx = pos % C::TX, ty = (pos / C::TX) % C::TY, tz = pos / (C::TX * C::TY);
      const bool live = ig < 2 * NCB * C::TPOS;
      s_zyx[r] = live ? ((tz + 1) << 16) | ((ty + 1) << 8) | (tx + 1) : -1;  // same origin convention as the halo items
      s_dst[r] = live ? C::XS_U4 + ob * 3 * C::TPOSP + pos : C::XS_U4 + C::TPOS;
      s_rel[r] = ob * 8 * (int)DHW + (tz + 1) * (int)HW + (ty + 1) * g.W + tx + 1;
    }
  }
  // the loads of a tile travel TWO phases ahead of the phase that splits them (a phase lasts 2-3 us, about one load latency under
  // load: one phase ahead left the latency half exposed — ablation: loads alone 126 us of a 206 us launch, 16->16 @96^3):
  // raw[j & 1] holds tile j; the tile loop is unrolled by two so that every index is a constant
  // (three cout blocks: 168 accumulator registers leave room for ONE tile of loads in flight, and the x fragments of a tap are not
  // double-buffered)
  float raw[2][PROUNDS][8] = {};
  auto tile_origin = [&](int tile, int& b, int& x0, int& y0, int& z0) {
    b = tile / tiles_per;
    const int bt = tile % tiles_per;
    x0 = (bt % g.ntx) * C::TX; y0 = ((bt / g.ntx) % g.nty) * C::TY; z0 = (bt / (g.ntx * g.nty)) * C::TZ;
  };
  auto load_tile = [&](int tile, auto SLOT) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT)::value % 2;
    if (g.dbg & 1) return;
    int b, x0, y0, z0;
    tile_origin(tile, b, x0, y0, z0);
    // buffer loads: the descriptor spans the channel planes of this block that exist in sample b, so an offset beyond them — a
    // ragged octet, or the 2^31 handed to out-of-volume and idle lanes — reads 0 in hardware: no address clamps, no selects
    // (extents clamped to the block's own 16 / 16 NCB channel planes: with all remaining channels the byte count could pass 2^31 —
    // the sentinel would then be IN range — or wrap at 2^32; the launcher admits D*H*W * 192 <= 2^31 only)
    const int xch = g.Cin - c0 < 16 ? g.Cin - c0 : 16, gch = g.Cout - co0 < 16 * NCB ? g.Cout - co0 : 16 * NCB;
    const icl_rsrc_t xr = icl_make_rsrc(x + (long)b * g.x_bstride + (long)c0 * DHW, (unsigned)((long)xch * DHW * 4));
    const icl_rsrc_t gr = icl_make_rsrc(gy + (long)b * g.gy_bstride + (long)co0 * DHW, (unsigned)((long)gch * DHW * 4));
    const int org = (z0 - 1) * (int)HW + (y0 - 1) * g.W + x0 - 1;
#pragma unroll
    for (int r = 0; r < PROUNDS; ++r) {
      const int gz = z0 - 1 + (s_zyx[r] >> 16), gyy = y0 - 1 + ((s_zyx[r] >> 8) & 255), gx = x0 - 1 + (s_zyx[r] & 255);
      const bool ok = (s_zyx[r] >= 0) & ((unsigned)gz < (unsigned)g.D) & ((unsigned)gyy < (unsigned)g.H) & ((unsigned)gx < (unsigned)g.W);
      unsigned off = ok ? (unsigned)(org + s_rel[r]) * 4u : 0x80000000u;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        raw[slot][r][c] = icl_buffer_load_f32(s_isx[r] ? xr : gr, off);
        off += (unsigned)DHW * 4u;
      }
    }
  };
  // one staging round: split the eight channel values of the lane's item into three packed planes, three 16-byte stores (idle
  // lanes store into a pad slot: no control flow)
  auto store_round = [&](uint4* buf, auto SLOT, int r) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT)::value % 2;
    uint4 o1, o2, o3;
    bf3_split8(raw[slot][r], o1, o2, o3);
    uint4* d = buf + s_dst[r];
    const int pitch = s_isx[r] ? C::NPOSP : C::TPOSP;
    d[0] = o1;
    d[pitch] = o2;
    d[2 * pitch] = o3;
  };
  auto store_tile = [&](uint4* buf, auto SLOT) __attribute__((always_inline)) {
    if (g.dbg & 2) return;
#pragma unroll
    for (int r = 0; r < PROUNDS; ++r) store_round(buf, SLOT, r);
  };

  const int t_begin = blockIdx.x * g.tiles_per_wg;
  const int t_end = t_begin + g.tiles_per_wg < g.ntiles ? t_begin + g.tiles_per_wg : g.ntiles;
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  if (producer) {
    // ============================================================================================ producer waves
    // (a separate code path that ends in `return`: the consumers' 56 NCB accumulator registers are not live here)
    ICL_SETPRIO(WSW_PRIO);
    if (t_begin < t_end) {
      load_tile(t_begin, I0());
      store_tile(lds, I0());
      if (t_begin + 1 < t_end) load_tile(t_begin + 1, I1());
      if (t_begin + 2 < t_end) load_tile(t_begin + 2, I0());
    }
    __syncthreads();
    auto pphase = [&](int tile, auto PAR) __attribute__((always_inline)) {
      constexpr int par = decltype(PAR)::value;
      typedef std::integral_constant<int, par ^ 1> OTHER;
      uint4* nxt = lds + (par ^ 1) * C::BUF_U4;
      const int phase_no = tile - t_begin;
      (void)phase_no;
      WSW_STAMP(0);
      if (tile + 1 < t_end) {
        store_tile(nxt, OTHER());           // split + store tile + 1 (requested two phases ago) into the other buffer
        WSW_STAMP(1);
        if (tile + 3 < t_end) load_tile(tile + 3, OTHER());      // request tile + 3 into the slot that has just been freed
        WSW_STAMP(2);
      }
      WSW_STAMP(3);
      __syncthreads();
      WSW_STAMP(4);
    };
    for (int tile = t_begin; tile < t_end; tile += 2) {
      pphase(tile, I0());
      if (tile + 1 < t_end) pphase(tile + 1, I1());
    }
    // the consumers' cross-wave sum: two barriers per (step, cout block)
#pragma unroll
    for (int q = 0; q < 2 * NCB; ++q) { __syncthreads(); __syncthreads(); }
    return;
  }
  // ============================================================================================== consumer waves
  // ---- operand addressing (bytes from the buffer base).  Lane 4q + p of a 16-lane group addresses position q of the group's four,
  // channels 4p .. 4p + 3: octet p >> 1, byte 8 (p & 1) of its 16-byte slot.  Group lg holds k = 8 lg .. 8 lg + 7 of the k-step:
  // tile row 2 kg + (lg >> 1), x = 8 (lg & 1) + 4 h + q for the two reads h of a fragment.
  const int rr = 2 * kg + (lg >> 1), rtz = rr / C::TY, rty = rr % C::TY, xq = 8 * (lg & 1) + lq;
  const int a_off = (C::XS_U4 + (lp >> 1) * 3 * C::TPOSP + rr * C::TX + xq) * 16 + (lp & 1) * 8;
  const int b_off = ((lp >> 1) * 3 * C::NPOSP + (rtz * C::PY + rty) * C::PX + xq) * 16 + (lp & 1) * 8;

  f32x4 acc[NCB][C::NTAPH];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int t = 0; t < C::NTAPH; ++t) acc[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto frag = [&](const unsigned char* p) __attribute__((always_inline)) {      // two transposing reads: k = 0..3 and 4..7 of the lane group (64 bytes apart)
    const uint2 lo = icl_lds_read_tr16_b64(p), hi = icl_lds_read_tr16_b64(p + 64);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  // TH = the wave's tap half as a compile-time constant (the two halves are two code paths: every tap offset is an immediate)
  auto multiply = [&](const uint4* buf, auto TH) __attribute__((always_inline)) {
    if (g.dbg & 4) return;
    constexpr int tap0 = C::NTAPH * decltype(TH)::value, ntap = decltype(TH)::value ? 27 - C::NTAPH : C::NTAPH;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(buf);
    uint4 a[NCB][3];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int s = 0; s < 3; ++s) a[cb][s] = frag(base + a_off + (cb * 6 + s) * C::TPOSP * 16);
    constexpr int NB = NCB < 3 ? 2 : 1;
    uint4 b[NB][3];
    auto read_b = [&](int buf_i, int t) __attribute__((always_inline)) {
      const int tap = tap0 + t;
      const unsigned char* p = base + b_off + (((tap / 9) * C::PY + (tap / 3) % 3) * C::PX + tap % 3) * 16;
#pragma unroll
      for (int s = 0; s < 3; ++s) b[buf_i][s] = frag(p + s * C::NPOSP * 16);
    };
    if (NB == 2) read_b(0, 0);
#pragma unroll
    for (int t = 0; t < ntap; ++t) {
      const int cur = NB == 2 ? (t & 1) : 0;
      if (NB == 1) read_b(0, t);
      else if (t + 1 < ntap) read_b(cur ^ 1, t + 1);     // next tap's fragments are in flight during this tap's MFMAs
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        // (dY split, x split) of the six terms, smallest first
        constexpr int sa[6] = {2, 1, 0, 1, 0, 0}, sb[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb][t] = icl_mfma_16x16x32_bf16(a[cb][sa[k]], b[cur][sb[k]], acc[cb][t]);
      }
    }
  };

  __syncthreads();
  auto phase = [&](int tile, auto PAR) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;
    uint4* cur = lds + par * C::BUF_U4;
    const int phase_no = tile - t_begin;
    (void)phase_no;
    WSW_STAMP(0);
    if (th == 0) multiply(cur, I0()); else multiply(cur, I1());
    WSW_STAMP(1);
    WSW_STAMP(3);
    __syncthreads();                                      // the other buffer is complete, this one has been read by everyone
    WSW_STAMP(4);
  };
  for (int tile = t_begin; tile < t_end; tile += 2) {
    phase(tile, I0());
    if (tile + 1 < t_end) phase(tile + 1, I1());
  }

  // ---- sum over the four k-groups (per tap half) through LDS, one cout block at a time: kg 2, 3 -> kg 0, 1; then kg 1 -> kg 0
  float* red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int step = 2; step >= 1; step >>= 1) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (kg >= step && kg < 2 * step) {
        float* d = red + (long)((kg - step) * 2 + th) * (C::NTAPH * 4 * 64) + lane;
#pragma unroll
        for (int t = 0; t < C::NTAPH; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) d[(t * 4 + r) * 64] = acc[cb][t][r];
      }
      __syncthreads();
      if (kg < step) {
        const float* d = red + (long)(kg * 2 + th) * (C::NTAPH * 4 * 64) + lane;
#pragma unroll
        for (int t = 0; t < C::NTAPH; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[cb][t][r] += d[(t * 4 + r) * 64];
      }
      __syncthreads();
    }
  }
  // D[row = cout 4 lg + r][col = cin li]: one float4 of four couts per (tap, cin)
  if (kg == 0 && c0 + li < g.CinP) {
    float* dst = gwp + (long)blockIdx.x * (27L * g.CinP * g.CoutP) + (long)(c0 + li) * g.CoutP + co0 + 4 * lg;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int t = 0; t < C::NTAPH; ++t) {
        const int tap = C::NTAPH * th + t;
        if (tap < 27 && co0 + cb * 16 < g.CoutP)
          *reinterpret_cast<float4*>(dst + (long)tap * g.CinP * g.CoutP + cb * 16) =
              make_float4(acc[cb][t][0], acc[cb][t][1], acc[cb][t][2], acc[cb][t][3]);
      }
  }
}

}  // namespace icl
