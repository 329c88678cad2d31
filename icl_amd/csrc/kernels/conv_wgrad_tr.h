// conv_wgrad_tr.h — 3x3x3 weight gradient on split products, second form (round 3): transposing LDS reads, double-buffered tiles,
// staggered wave halves.
//
//   dW[co][ci][tap] = sum over voxels p of dY[co][p] * x[ci][p + tap - 1]
//
// GEMM per tap as in conv_bf16x3.h's first form: rows = 16 output channels, columns = 16 input channels, k = 32 voxels (two (z, y)
// rows x 16 x), fp32 operands split exactly into three bf16 terms, six MFMA terms per product.  What changed, and why
// (profiles/r2_pmc_conv.md: the first form keeps the matrix pipe 28-31 % busy — one wave per SIMD, 3.0-3.9 VALU per MFMA):
//  * LDS images are POSITION-major, exactly the forward kernel's: Xs[cin octet][split][halo position] and Gs[cout octet][split][tile
//    position] of 8 packed bf16 (16 bytes), written by conflict-free ds_write_b128 (a thread owns one position of 8 channel planes).
//    The MFMA wants 8 consecutive VOXELS of one channel per lane; ds_read_b64_tr_b16 delivers exactly that from a position-major
//    image (a 4-position x 16-channel block per 16 lanes, transposed in the LDS crossbar).  A tap shift is an address offset: no
//    v_alignbit, no edge words, no VALU at all between the MFMAs.
//  * Eight waves, two per SIMD: wave (kg, th) owns k-step kg of the tile and the tap half th (taps 0..13 / 14..26): 56 NCB accumulator
//    registers instead of 108 NCB, so NCB = 2 fits in 256 registers with two waves per SIMD.
//  * Tiles of 2 x 4 x 16 voxels in TWO LDS buffers, ONE barrier per tile.  Between two barriers a wave multiplies tile i out of
//    buffer i & 1 (M) and splits + stores its share of tile i + 1 into the other buffer (S); the loads of tile i + 2 are in flight
//    meanwhile.  Waves 0-3 run S then M, waves 4-7 M then S: the two waves of a SIMD (w and w + 4) are in opposite phases, so the
//    split VALU / LDS writes of one run beside the MFMAs of the other instead of all eight waves staging in lockstep.
// Accumulators are summed over the four k-groups through LDS at the end; one packed slab per workgroup, reduced in a fixed order
// by reduce_unpack_wgrad_kernel as for the other weight-gradient kernels.
// Reference op: the weight gradient of nn.Conv3d(k=3, pad=1) in UnetConv3 (/root/reference/code/networks/utils.py:104,107).
#pragma once

namespace icl {

template <int NCB_>
struct WgTrT {
  static constexpr int NCB = NCB_, TZ = 2, TY = 4, TX = 16, PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
  static constexpr int NPOS = PZ * PY * PX, TPOS = TZ * TY * TX;          // 432 halo positions, 128 tile positions
  // image pitches with pitch % 16 == 4: the two channel octets a transposing read touches are 3 planes apart; 3 * pitch * 16 bytes
  // must be an odd multiple of 64 (mod 256) so that the octets fall into complementary bank quarters (no conflicts)
  static constexpr int NPOSP = NPOS + 4, TPOSP = TPOS + 4;
  static_assert(NPOSP % 16 == 4 && TPOSP % 16 == 4, "bank layout of the transposing reads");
  static constexpr int NT = 512;
  static constexpr int XS_U4 = 6 * NPOSP, GS_U4 = 6 * NCB * TPOSP, BUF_U4 = XS_U4 + GS_U4;
  static constexpr size_t LDS_BYTES = (size_t)2 * BUF_U4 * 16;
  // staging items: [0, 2 NPOS) halo positions x cin octets, idle up to XPAD (a multiple of 64: the kind of an item is wave-uniform),
  // then 2 NCB TPOS tile positions x cout octets
  static constexpr int XITEMS = 2 * NPOS, XPAD = (XITEMS + 63) / 64 * 64, ITEMS = XPAD + 2 * NCB * TPOS;
  static constexpr int ROUNDS = (ITEMS + NT - 1) / NT;
  static constexpr int PF = NCB_ < 3 ? 2 : 1;                              // tiles of loads in flight ahead of the split (registers)
  static constexpr int NTAPH = 14;                                        // taps per wave: half 0 owns 0..13, half 1 owns 14..26
  static constexpr int ACC = NTAPH * NCB;
  static_assert((size_t)2 * 2 * NTAPH * 4 * 64 * 4 <= LDS_BYTES, "the cross-wave sum (two writers per tap half, one cout block) must fit in the tile buffers");
};

// Measured and dropped (round 3): the split + store of tile i + 1 spread over the tap loop of the multiply phase (one staging round
// every four taps — a wave's phase as ONE merged instruction stream instead of the chain S, M): 5-14 % slower on the one-cout-block
// layers (16->16 @96^3 181 vs 167 us, 48->16 546 vs 479), 1-4 % slower with two cout blocks — as for the forward kernel, VALU that
// the compiler interleaves with a wave's own MFMAs costs more than the same VALU run as a block beside the partner wave's MFMAs.
// Also measured and dropped: dedicated producer / consumer waves for one cout block (waves 0-7 only multiply, 4 or 8 more waves
// only stage; 126 registers, three / four waves per SIMD): 4 producers 11-12 % slower (the staging of a tile becomes the longer
// side), 8 producers within 1 % of this kernel (16->16 @96^3 166 vs 168 us) — with one cout block the LDS is the co-bottleneck
// (one transposing read per MFMA: ~55 % of its bandwidth, plus the staging writes), not the order of S and M inside a wave.
// Also measured and dropped (round 3, after in-kernel stamps had priced a vector-memory instruction of the forward kernel at ~70
// cycles of the issuing wave): staging by 16-byte loads — items of four x positions x four channels (four aligned buffer_load_dwordx4,
// 44 wave-level load instructions per tile instead of 192), three ds_write_b64 per position with a lane-dependent rotation of the
// position order for conflict-free banks: 16->16 @96^3 166 vs 166 us, 32->32 @48^3 84 vs 77, 96->32 202 vs 180, 48->48 @96^3 1539 vs
// 1299 (spills).  The number of load instructions is not what bounds this kernel.  Neither is the placement of the transposing
// reads: woven one behind every MFMA of the previous tap by scheduling groups (the forward kernel's -7..13 %) 16->16 @96^3 205 / 199 vs
// 202 / 201 us, 32->32 @48^3 88 vs 84-85 on one box.  (hipcc 7.2 note from that experiment: casting the
// elements of __builtin_amdgcn_raw_buffer_load_b128's result one by one narrows the load to one dword; cast the whole vector.)
// In-kernel stamps of round 3 (-DWGTR_DEBUG, tools/wgtr_stamps.py; 16->16 @96^3, a phase = one tile = ~6.0 k cycles, 1.3 k of them the
// wave's 81 MFMAs): S -> M wave: split + store 2.1 k, issuing the 24 loads 1.5 k (~55 cycles per vector-memory instruction), multiply
// 2.3 k; M -> S wave: multiply 2.6 k, split + store 1.1-1.6 k, loads 1.3 k, barrier wait 0.4-1.1 k.  Each wave's phase is the serial chain
// S + L + M; the partner covers about half of it.  Tried on top and dropped: the S -> M wave's loads woven between the taps of its
// multiply phase (multiply 2.3 -> 4.4 k cycles: 16->16 200 -> 214 us); three tap groups of nine taps — twelve waves, three per SIMD at
// 144 registers, rotated S L M / M S L / S M L so that exactly one wave of a SIMD multiplies at any time — 16->16 @96^3 166.1 vs 166.4
// us, 48->16 474 vs 488: a third wave per SIMD does not shorten the phase either.  What the stamps suggest instead: a wave's split +
// store takes 2.1 k cycles beside its partner's MFMAs and 1.1-1.6 k beside its loads — VALU and MFMA instructions of a SIMD's waves
// share one issue pipe, so the ~200 VALU instructions per wave and tile (two thirds of them the operand split) are paid on top of the
// 81 MFMAs, not beside them.  Fewer VALU instructions per staged value (planes written by the producer) is the lever, not the schedule.
#if defined(WGTR_DEBUG)
// waves 0 and 4 of workgroup 0 (one SIMD: the S -> M and the M -> S wave of k-group 0) record s_memtime at the boundaries of phases
// 4..7; read back with icl_debug_wgtr_stamps
__device__ long long g_wgtr_stamps[2 * 4 * 8];
#define WGTR_STAMP(k)                                                                                        \
  do {                                                                                                       \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (wid & 3) == 0 && lane == 0 && phase_no >= 4 && phase_no < 8) \
      g_wgtr_stamps[((wid >> 2) * 4 + phase_no - 4) * 8 + (k)] = clock64();                                  \
  } while (0)
#else
#define WGTR_STAMP(k) ((void)0)
#endif

template <int NCB>
__global__ __launch_bounds__(512) void conv3d_wgrad_tr_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                              float* __restrict__ gwp, Bf3WGeom g) {
  typedef WgTrT<NCB> C;
  ICL_DYN_LDS(uint4, lds);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int kg = wid & 3, th = wid >> 2;                 // k-step of the tile, tap half
  const int lg = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int ncb = (g.CinP + 15) / 16;
  const int co0 = (blockIdx.y / ncb) * 16 * NCB, c0 = (blockIdx.y % ncb) * 16;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const int tiles_per = g.ntz * g.nty * g.ntx;

  // ---- staging tables (tile-invariant).  Item it = tid + r * NT: kind (halo x / tile dY) is uniform per (wave, round).
  int s_zyx[C::ROUNDS], s_dst[C::ROUNDS], s_rel[C::ROUNDS];
  bool s_isx[C::ROUNDS];
#pragma unroll
  for (int r = 0; r < C::ROUNDS; ++r) {
    const int it = tid + r * C::NT;
    int isx = (wid * 64 + r * C::NT) < C::XPAD;
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
  float raw[C::PF][C::ROUNDS][8] = {};
  auto tile_origin = [&](int tile, int& b, int& x0, int& y0, int& z0) {
    b = tile / tiles_per;
    const int bt = tile % tiles_per;
    x0 = (bt % g.ntx) * C::TX; y0 = ((bt / g.ntx) % g.nty) * C::TY; z0 = (bt / (g.ntx * g.nty)) * C::TZ;
  };
  auto load_tile = [&](int tile, auto SLOT) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT)::value % C::PF;
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
    for (int r = 0; r < C::ROUNDS; ++r) {
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
    constexpr int slot = decltype(SLOT)::value % C::PF;
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
    for (int r = 0; r < C::ROUNDS; ++r) store_round(buf, SLOT, r);
  };

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

  const int t_begin = blockIdx.x * g.tiles_per_wg;
  const int t_end = t_begin + g.tiles_per_wg < g.ntiles ? t_begin + g.tiles_per_wg : g.ntiles;
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  // prologue: tile 0 into buffer 0; the loads of tiles 1 and 2 in flight
  if (t_begin < t_end) {
    load_tile(t_begin, I0());
    store_tile(lds, I0());
    if (t_begin + 1 < t_end) load_tile(t_begin + 1, I1());
    if (C::PF == 2 && t_begin + 2 < t_end) load_tile(t_begin + 2, I0());
  }
  __syncthreads();
  // one phase: multiply tile `tile` (relative parity PAR: buffer PAR), split + store tile + 1 (raw slot PAR ^ 1) into the other
  // buffer, issue the loads of tile + 3 into the slot that the store has just freed
  auto phase = [&](int tile, auto PAR) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;
    typedef std::integral_constant<int, par ^ 1> OTHER;
    uint4* cur = lds + par * C::BUF_U4;
    uint4* nxt = lds + (par ^ 1) * C::BUF_U4;
    const bool more = tile + 1 < t_end;
    const int phase_no = tile - t_begin;
    (void)phase_no;
    WGTR_STAMP(0);
    if (th == 0) {
      if (more) {
        store_tile(nxt, OTHER());
        WGTR_STAMP(1);
        if (tile + 1 + C::PF < t_end) load_tile(tile + 1 + C::PF, OTHER());
        WGTR_STAMP(2);
      }
      multiply(cur, I0());
      WGTR_STAMP(3);
    } else {
      multiply(cur, I1());
      WGTR_STAMP(1);
      if (more) {
        store_tile(nxt, OTHER());
        WGTR_STAMP(2);
        if (tile + 1 + C::PF < t_end) load_tile(tile + 1 + C::PF, OTHER());
        WGTR_STAMP(3);
      }
    }
    __syncthreads();                                      // nxt is complete, cur has been read by everyone
    WGTR_STAMP(4);
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
