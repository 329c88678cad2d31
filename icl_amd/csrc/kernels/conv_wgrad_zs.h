// conv_wgrad_zs.h — 3x3x3 weight gradient on split products, third form (round 5): a workgroup walks a COLUMN of 1 x 8 x 16 tiles down
// z and keeps the halo planes it has already staged.
//
//   dW[co][ci][tap] = sum over voxels p of dY[co][p] * x[ci][p + tap - 1]
//
// Same product as conv_wgrad_tr.h (rows = 16 output channels, columns = 16 input channels, k = 32 voxels, three exact bf16 terms per
// fp32 operand, six MFMA terms per product, position-major LDS images read by ds_read_b64_tr_b16, eight waves = four k-steps x two tap
// halves, S-then-M / M-then-S wave halves).  What changed, and why: round 5's ablations of the forward kernel
// (profiles/r5_cl16_stage_a.md) and round 3's stamps of this one (conv_wgrad_tr.h: "fewer VALU instructions per staged value is the
// lever, not the schedule") say the vector instructions of the operand staging — 11 per pair of values for the exact three-way split —
// are what the matrix pipe waits for, and a 2 x 4 x 16 tile stages 4 x 6 x 18 = 432 halo positions of x for its 128 voxels (3.4x) in
// every phase although the tile before it in the walk had just staged most of them.  Here the tile is ONE z-plane of 8 x 16 voxels and
// consecutive tiles of a workgroup are consecutive in z: the three halo planes z - 1, z, z + 1 of a tile live in a ring of FOUR plane
// slots in LDS, and a phase stages only the one plane the next tile adds — 10 x 18 = 180 positions per 128 voxels (1.4x).  Per phase
// and (cout group, cin block): 180 + 128 NCB staged positions instead of 432 + 128 NCB; the loads, the split VALU and the LDS stores
// all shrink by that factor, the multiply is the same 81 NCB MFMAs per wave.
//  * Ring: plane slot s holds positions [180 s, 180 s + 180) of every (octet, split) image.  A tile with ring base rb reads its planes
//    dz = 0, 1, 2 from slots (rb + dz) & 3 — one base register per dz, the (dy, dx) part of a tap stays an immediate offset — while the
//    plane z + 2 is written to slot (rb + 3) & 3; after the phase's barrier rb advances by one.
//  * The first tile of a column (and of a workgroup's run) has nothing to inherit: its planes z - 1 and z are staged on their own
//    (`stage_extra`), behind the barrier, because the slots they go to are being read until then.
//  * dY tiles are double-buffered as before (two buffers of 128 positions).
// Tiles are numbered z-fastest, a workgroup owns a contiguous run (tiles_per_wg), and — when the grid is a multiple of 8 — the runs
// of one XCD are adjacent, so neighbouring columns (which share the cache lines of their x halos) meet in one L2.
// One packed slab per workgroup, reduced in a fixed order by reduce_unpack_wgrad_kernel like the other weight-gradient kernels.
// Reference op: the weight gradient of nn.Conv3d(k=3, pad=1) in UnetConv3 (/root/reference/code/networks/utils.py:104,107).
//
// Measured (round 5, batch 2, tools/probe/conv_wgrad_zs_probe.hip, against conv3d_wgrad_tr_kernel; profiles/r5_wgrad_zs_probe_*.txt):
// 16->16 @96^3 148 vs 170 us, 48->16 436 vs 520, 32->16 282 vs 341, 48->48 (three cout blocks) 1038 vs 1314; 96->32 @48^3 183 vs 194,
// 32->32 77 vs 76, 64->64 @24^3 50 vs 51.  Matrix pipe busy 40 -> 52 % (16->16 @96^3), 49 -> 50 % (32->32 @48^3); vector instructions
// 27.7 -> 18.2 M per launch.  A phase is ~5.1 k cycles for 2.6 k of matrix-pipe time; multiply-only builds keep the pipe 93 % busy,
// staging-only builds need 2.4 k per phase, and the two add up — also with four dedicated staging waves (12 waves, 159 us), at any
// staging priority.  With operands that arrive pre-split (three 16-byte loads + three LDS stores per item, no vector arithmetic:
// profiles/r5_wgrad_zs_probe_planes.txt) the kernel is only 1.09-1.17x faster: what is left beside the MFMAs is the ~60-100 cycles a
// vector-memory instruction costs its SIMD there, not the split.  Dropped on the way: scalar-offset loads with the cursor advanced inside
// the branch on the wave's role (the compiler made every load a readfirstlane loop: 175 us), a wave-uniform role (scalar-branch code
// paths: 137 / 728 spilled registers with two / three cout blocks), all prologue loads in flight together (no change).
#pragma once

namespace icl {

template <int NCB_, int NW_ = 8>
struct WgZsT {
  static constexpr int NCB = NCB_, NW = NW_, TY = 8, TX = 16, PY = TY + 2, PX = TX + 2;
  static constexpr int PLANE = PY * PX, NSLOT = 4, NPOS = NSLOT * PLANE, TPOS = TY * TX;      // 180, 4, 720, 128
  // image pitches with pitch % 16 == 4 (conv_wgrad_tr.h: the two channel octets of a transposing read fall into complementary bank quarters)
  static constexpr int NPOSP = NPOS + 4, TPOSP = TPOS + 4;
  static_assert(NPOSP % 16 == 4 && TPOSP % 16 == 4, "bank layout of the transposing reads");
  // NW waves: 0..7 multiply (four k-steps x two tap halves) and stage; waves 8.. (NW = 10, one cout block) only stage — 360 + 256 items
  // are 1.25 rounds of 512 threads and exactly one round of 640
  static constexpr int NT = 64 * NW;
  static constexpr int XS_U4 = 6 * NPOSP, GS_U4 = 6 * NCB * TPOSP;
  static constexpr size_t LDS_BYTES = (size_t)(XS_U4 + 2 * GS_U4) * 16;
  // staging items of a phase: [0, 2 PLANE) positions of the new halo plane x cin octets, idle up to XPAD (a multiple of 64: the kind of
  // an item is wave-uniform), then 2 NCB TPOS tile positions x cout octets
  static constexpr int XITEMS = 2 * PLANE, XPAD = (XITEMS + 63) / 64 * 64, ITEMS = XPAD + 2 * NCB * TPOS;
  static constexpr int ROUNDS = (ITEMS + NT - 1) / NT;
  static constexpr int EITEMS = 2 * XITEMS, EROUNDS = (EITEMS + NT - 1) / NT;                  // the two extra planes of a column's first tile
  static constexpr int PF = NCB_ < 3 ? 2 : 1;                                                 // tiles of loads in flight ahead of the split
  static constexpr int NTAPH = 14;
  static constexpr int ACC = NTAPH * NCB;
  static_assert((size_t)2 * 2 * NTAPH * 4 * 64 * 4 <= LDS_BYTES, "the cross-wave sum must fit in the tile buffers");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

#if defined(WGZS_DEBUG)
// waves 0 and 4 of workgroup 0 (one SIMD: the S -> M and the M -> S wave of k-group 0) record the clock at the boundaries of phases 4..7
__device__ long long g_wgzs_stamps[2 * 4 * 8];
#define WGZS_STAMP(k)                                                                                        \
  do {                                                                                                       \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (wid == 0 || wid == 4) && lane == 0 && phase_no >= 4 && phase_no < 8) \
      g_wgzs_stamps[((wid >> 2) * 4 + phase_no - 4) * 8 + (k)] = clock64();                                  \
  } while (0)
#else
#define WGZS_STAMP(k) ((void)0)
#endif

template <int NCB, int NW = 8>
__global__ __launch_bounds__(64 * NW) void conv3d_wgrad_zs_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                  float* __restrict__ gwp, Bf3WGeom g) {
  typedef WgZsT<NCB, NW> C;
  ICL_DYN_LDS(uint4, lds);
  const int tid = threadIdx.x, lane = tid & 63;
  // (wid stays a vector value: with a wave-uniform wid hipcc 7.2 turns the two wave halves into scalar-branch code paths and spills 137 / 728
  // registers with two / three cout blocks)
  const int wid = tid >> 6;
  const int kg = wid & 3, th = wid >> 2;                 // k-step of the tile, tap half (th == 2: a wave that only stages)
  const int lg = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int ncb = (g.CinP + 15) / 16;
  const int co0 = (blockIdx.y / ncb) * 16 * NCB, c0 = (blockIdx.y % ncb) * 16;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const int cols_per = g.nty * g.ntx;                    // columns per sample; g.ntz = D tiles per column
  // workgroup -> run of tiles: the runs of one XCD (blockIdx.x & 7) are adjacent
  const int wgl = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_begin = wgl * g.tiles_per_wg;
  const int t_end = t_begin + g.tiles_per_wg < g.ntiles ? t_begin + g.tiles_per_wg : g.ntiles;

  // ---- staging items.  Item it = tid + r * NT: kind (halo x / tile dY) and "this wave has items in round r" are uniform per (wave,
  // round).  What stays live per round: the LDS slot and — per COLUMN — the item's byte offset from the start of the column's z = 0 plane
  // (or 2^31 for a lane that must read zeros: outside the volume in y / x, idle); the z part of the address and the channel stride are
  // wave-uniform and travel in the load's scalar offset.  A phase's loads cost no vector instruction beside themselves.
  int s_dst[C::ROUNDS];
  unsigned coloff[C::ROUNDS];
  bool s_isx[C::ROUNDS], s_act[C::ROUNDS];
#pragma unroll
  for (int r = 0; r < C::ROUNDS; ++r) {
    const int it = tid + r * C::NT;
    int isx = (wid * 64 + r * C::NT) < C::XPAD, act = (wid * 64 + r * C::NT) < C::ITEMS;
    ICL_WAVE_UNIFORM(isx);
    ICL_WAVE_UNIFORM(act);
    s_isx[r] = isx != 0;
    s_act[r] = act != 0;
    coloff[r] = 0x80000000u;
    if (isx) s_dst[r] = it < C::XITEMS ? (it / C::PLANE) * 3 * C::NPOSP + it % C::PLANE : -1;          // + ring slot * PLANE
    else {
      const int ig = it - C::XPAD;
      s_dst[r] = ig < 2 * NCB * C::TPOS ? C::XS_U4 + (ig / C::TPOS) * 3 * C::TPOSP + ig % C::TPOS : -1;   // + buffer parity * GS_U4
    }
  }
  // the load cursor: the tile whose new plane and dY tile are requested next (tiles are requested in order, each once)
  int lz = 0, lcx = 0, lcy = 0, lb = 0;
  icl_rsrc_t xr = icl_make_rsrc(x, 0u), gr = icl_make_rsrc(gy, 0u);
  auto col_setup = [&]() __attribute__((always_inline)) {
    // buffer loads with hardware zero fill (conv_wgrad_tr.h: the descriptor spans the channel planes of this block in sample lb)
    const int xch = g.Cin - c0 < 16 ? g.Cin - c0 : 16, gch = g.Cout - co0 < 16 * NCB ? g.Cout - co0 : 16 * NCB;
    xr = icl_make_rsrc(x + (long)lb * g.x_bstride + (long)c0 * DHW, (unsigned)((long)xch * DHW * 4));
    gr = icl_make_rsrc(gy + (long)lb * g.gy_bstride + (long)co0 * DHW, (unsigned)((long)gch * DHW * 4));
    const int x0 = lcx * C::TX, y0 = lcy * C::TY;
    int t = tid;
    ICL_OPAQUE_INT(t);                                   // (recomputed per column on purpose: nothing of this stays live across the phases)
#pragma unroll
    for (int r = 0; r < C::ROUNDS; ++r) {
      const int it = t + r * C::NT;
      if (s_isx[r]) {
        const int o = it / C::PLANE, pos = it % C::PLANE, gyy = y0 - 1 + pos / C::PX, gx = x0 - 1 + pos % C::PX;
        const bool ok = (it < C::XITEMS) & ((unsigned)gyy < (unsigned)g.H) & ((unsigned)gx < (unsigned)g.W);
        coloff[r] = ok ? (unsigned)(o * 8 * (int)DHW + gyy * g.W + gx) * 4u : 0x80000000u;
      } else {
        const int ig = it - C::XPAD, ob = ig / C::TPOS, pos = ig % C::TPOS, gyy = y0 + pos / C::TX, gx = x0 + pos % C::TX;
        const bool ok = (ig < 2 * NCB * C::TPOS) & (gyy < g.H) & (gx < g.W);
        coloff[r] = ok ? (unsigned)(ob * 8 * (int)DHW + gyy * g.W + gx) * 4u : 0x80000000u;
      }
    }
  };
  f32x4 acc[NCB][C::NTAPH];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int t = 0; t < C::NTAPH; ++t) acc[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  float raw[C::PF][C::ROUNDS][8] = {};
  // the NEW halo plane of the cursor's tile (global z = lz + 1) and its dY tile: loads travel PF phases ahead of the phase that splits
  // them.  The cursor itself only moves in wave-uniform code (`advance`, behind the phase's barrier): a value that changes inside the
  // branch on the wave's tap half is a vector value for the compiler, and a buffer load whose descriptor or scalar offset is one
  // becomes a readfirstlane loop (measured: 175 instead of 151 us on 16->16 @96^3)
  auto load_new = [&](auto SLOT, int z) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT)::value % C::PF;
    if (g.dbg & 1) return;
    const bool zok = z + 1 < g.D;
    const unsigned zx = (unsigned)((z + 1) * (int)HW) * 4u, zg = (unsigned)(z * (int)HW) * 4u;
    // (the channel stride stays in the LANE offset: the hardware range-checks the lane offset only, and a ragged channel block relies
    // on that check — its missing channels lie beyond the descriptor's extent and must read zeros)
#pragma unroll
    for (int r = 0; r < C::ROUNDS; ++r) {
      if (!s_act[r]) continue;
      unsigned off = coloff[r];
      if (s_isx[r]) {
        if (zok) {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            raw[slot][r][c] = icl_buffer_load_f32(xr, off, zx);
            off += (unsigned)DHW * 4u;
          }
        } else {                                         // the plane below the volume: zeros
#pragma unroll
          for (int c = 0; c < 8; ++c) raw[slot][r][c] = 0.f;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          raw[slot][r][c] = icl_buffer_load_f32(gr, off, zg);
          off += (unsigned)DHW * 4u;
        }
      }
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    if (++lz == g.D) {                                     // next column
      lz = 0;
      if (++lcx == g.ntx) {
        lcx = 0;
        if (++lcy == g.nty) { lcy = 0; ++lb; }
      }
      col_setup();
    }
  };
  // split the eight channel values of every item into three packed planes: three 16-byte stores per item
  auto store_new = [&](int xslot, int par, auto SLOT) __attribute__((always_inline)) {
    constexpr int slot = decltype(SLOT)::value % C::PF;
    if (g.dbg & 2) return;
#pragma unroll
    for (int r = 0; r < C::ROUNDS; ++r) {
      if (!s_act[r]) continue;
      uint4 o1, o2, o3;
      bf3_split8(raw[slot][r], o1, o2, o3);
      if (s_dst[r] >= 0) {
        uint4* d = lds + s_dst[r] + (s_isx[r] ? xslot * C::PLANE : par * C::GS_U4);
        const int pitch = s_isx[r] ? C::NPOSP : C::TPOSP;
        d[0] = o1;
        d[pitch] = o2;
        d[2 * pitch] = o3;
      }
    }
  };
  // the halo planes z0 - 1 and z0 of a column's first tile, into ring slots rb and rb + 1 (loaded, split and stored on the spot)
  auto stage_extra = [&](int tile, int rb) __attribute__((always_inline)) {
    const int col = tile / g.ntz, z0 = tile - col * g.ntz, b = col / cols_per, cc = col - b * cols_per;
    const int x0 = (cc % g.ntx) * C::TX, y0 = (cc / g.ntx) * C::TY;
    const int xch = g.Cin - c0 < 16 ? g.Cin - c0 : 16;
    const icl_rsrc_t er = icl_make_rsrc(x + (long)b * g.x_bstride + (long)c0 * DHW, (unsigned)((long)xch * DHW * 4));
    // (a rolled loop on purpose: this runs once per column — its index arithmetic and eight staging registers must not stay live across
    // the phases)
#pragma unroll 1
    for (int e = tid; e < C::EROUNDS * C::NT; e += C::NT) {
      const int plane = e / C::XITEMS, rem = e % C::XITEMS, o = rem / C::PLANE, pos = rem % C::PLANE;
      const int gz = z0 - 1 + plane, gyy = y0 - 1 + pos / C::PX, gx = x0 - 1 + pos % C::PX;
      const bool ok = (e < C::EITEMS) & ((unsigned)gz < (unsigned)g.D) & ((unsigned)gyy < (unsigned)g.H) & ((unsigned)gx < (unsigned)g.W);
      const unsigned off = ok ? (unsigned)(o * 8 * (int)DHW + gz * (int)HW + gyy * g.W + gx) * 4u : 0x80000000u;
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = (g.dbg & 1) ? 0.f : icl_buffer_load_f32(er, off + (unsigned)c * (unsigned)DHW * 4u);
      uint4 o1, o2, o3;
      bf3_split8(v, o1, o2, o3);
      if (e < C::EITEMS) {
        uint4* d = lds + o * 3 * C::NPOSP + ((rb + plane) & 3) * C::PLANE + pos;
        d[0] = o1;
        d[C::NPOSP] = o2;
        d[2 * C::NPOSP] = o3;
      }
    }
  };

  // ---- operand addressing (bytes from the LDS base), as in conv_wgrad_tr.h: lane 4q + p of a 16-lane group addresses position q of the
  // group's four, channels 4p .. 4p + 3.  Group lg holds k = 8 lg .. 8 lg + 7 of the k-step: tile row 2 kg + (lg >> 1),
  // x = 8 (lg & 1) + 4 h + q for the two reads h of a fragment.
  const int rr = 2 * kg + (lg >> 1), xq = 8 * (lg & 1) + lq;
  const int a_off = (C::XS_U4 + (lp >> 1) * 3 * C::TPOSP + rr * C::TX + xq) * 16 + (lp & 1) * 8;
  const int b_off = ((lp >> 1) * 3 * C::NPOSP + rr * C::PX + xq) * 16 + (lp & 1) * 8;

  auto frag = [&](const unsigned char* p) __attribute__((always_inline)) {      // two transposing reads: k = 0..3 and 4..7 of the lane group (64 bytes apart)
    const uint2 lo = icl_lds_read_tr16_b64(p), hi = icl_lds_read_tr16_b64(p + 64);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  // TH = the wave's tap half as a compile-time constant; PAR = the dY buffer; rb = ring base of this tile (wave-uniform)
  auto multiply = [&](auto PAR, int rb, auto TH) __attribute__((always_inline)) {
    if (g.dbg & 4) return;
    constexpr int tap0 = C::NTAPH * decltype(TH)::value, ntap = decltype(TH)::value ? 27 - C::NTAPH : C::NTAPH;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(lds);
    uint4 a[NCB][3];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int s = 0; s < 3; ++s) a[cb][s] = frag(base + a_off + (decltype(PAR)::value * C::GS_U4 + (cb * 6 + s) * C::TPOSP) * 16);
    const unsigned char* pz[3];
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) pz[dz] = base + b_off + ((rb + dz) & 3) * (C::PLANE * 16);
    constexpr int NB = NCB < 3 ? 2 : 1;
    uint4 b[NB][3];
    auto read_b = [&](int buf_i, int t) __attribute__((always_inline)) {
      const int tap = tap0 + t;
      const unsigned char* p = pz[tap / 9] + (((tap / 3) % 3) * C::PX + tap % 3) * 16;
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

  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  int rb = 0, mz = 0;                                     // ring base and z of the tile being multiplied
  // prologue: the first tile's three planes and dY tile; the loads of the next PF tiles in flight
  if (t_begin < t_end) {
    const int col = t_begin / g.ntz;
    mz = lz = t_begin - col * g.ntz;
    lb = col / cols_per;
    lcx = (col - lb * cols_per) % g.ntx; lcy = (col - lb * cols_per) / g.ntx;
    col_setup();
    load_new(I0(), lz);
    advance();
    stage_extra(t_begin, rb);
    store_new((rb + 2) & 3, 0, I0());
    if (t_begin + 1 < t_end) { load_new(I1(), lz); advance(); }
    if (C::PF == 2 && t_begin + 2 < t_end) { load_new(I0(), lz); advance(); }
  }
  __syncthreads();
  // one phase: multiply tile `tile` (dY buffer PAR, planes at ring slots rb .. rb + 2), split + store the new plane and the dY tile of
  // tile + 1 (raw slot PAR ^ 1) into ring slot rb + 3 / dY buffer PAR ^ 1, issue the loads of tile + 1 + PF into the freed raw slot
  auto phase = [&](int tile, auto PAR) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;
    typedef std::integral_constant<int, par ^ 1> OTHER;
    const bool more = tile + 1 < t_end, ld = tile + 1 + C::PF < t_end;
    const int phase_no = tile - t_begin;
    (void)phase_no;
    WGZS_STAMP(0);
    const int thu = th;
    if (thu == 1) {
      multiply(PAR, rb, I1());
      WGZS_STAMP(1);
      if (more) {
        store_new((rb + 3) & 3, par ^ 1, OTHER());
        WGZS_STAMP(2);
        if (ld) load_new(OTHER(), lz);
        WGZS_STAMP(3);
      }
    } else {
      if (more) {
        store_new((rb + 3) & 3, par ^ 1, OTHER());
        WGZS_STAMP(1);
        if (ld) load_new(OTHER(), lz);
        WGZS_STAMP(2);
      }
      if (thu == 0) multiply(PAR, rb, I0());
      WGZS_STAMP(3);
    }
    __syncthreads();                                      // the new plane and dY tile are complete, this tile has been read by everyone
    WGZS_STAMP(4);
    if (ld) advance();
    rb = (rb + 1) & 3;
    if (++mz == g.D) {                                    // tile + 1 starts a column: its planes z - 1 (zeros) and z replace the old column's
      mz = 0;
      if (more) {
        stage_extra(tile + 1, rb);
        __syncthreads();
      }
    }
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
      if (th < 2 && kg >= step && kg < 2 * step) {
        float* d = red + (long)((kg - step) * 2 + th) * (C::NTAPH * 4 * 64) + lane;
#pragma unroll
        for (int t = 0; t < C::NTAPH; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) d[(t * 4 + r) * 64] = acc[cb][t][r];
      }
      __syncthreads();
      if (th < 2 && kg < step) {
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
  if (th < 2 && kg == 0 && c0 + li < g.CinP) {
    float* dst = gwp + (long)wgl * (27L * g.CinP * g.CoutP) + (long)(c0 + li) * g.CoutP + co0 + 4 * lg;
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
