// conv_cl16.h — the split-product 3x3x3 convolution on BLOCKED CHANNELS-LAST activations (round 5).
//
// Layout "CL16": [sample][C / 16][D][H][W][16] fp32 — the 16 channels of a voxel are 64 contiguous bytes, a halo row of 18 voxels is
// 1,152 contiguous bytes.  Why (profiles/r4_planes_dma.md, r4_halo_fetch_probe.txt): on NCDHW the halo tile of a work item is 72
// one-dword gathers per loader lane that use 28 % of the cache lines they touch, and the loader chain (issue 6.5 k cycles -> data ->
// split -> deposit) is longer than the item's multiply phase; here it is 18 sixteen-byte loads per lane on fully used lines
// (stand-alone fetch of a tile: 1.4 us against 2.9).
//
// Normalise-on-load (reference: Conv3d -> InstanceNorm3d -> ReLU, /root/reference/code/networks/utils.py:104-109): a convolution of the
// backbone stores its RAW output together with the InstanceNorm statistics of it (epilogue, as conv_bf16x3.h); the consumer applies
//     a = max(fma(y, scale, shift), 0),   scale = rstd, shift = -mean * rstd   per (sample, channel)
// to every value while it stages it (2 VALU per value on the loader waves), so the normalised tensor is never written or re-read.
// A source with ss == nullptr is used as stored (clean activations, gradients).  Up to two sources form a channel concatenation
// (skip connection first, up-sampled map second: torch.cat([skip, up], 1), utils.py:276) without a copy.
//
// The multiply loop, the LDS images, the weight planes and the order of every floating-point sum are those of conv_bf16x3_ws.h; the
// MFMA operands are swapped (A = weights, B = activations), so a lane's accumulator holds four consecutive output CHANNELS of one voxel
// and the epilogue stores 16-byte channel quads: one store instruction of a wave writes 16 voxels x 64 bytes = 1 KB contiguous.
#pragma once

namespace icl {

struct ClSrc {
  const float* p;        // [n][nchunks][D][H][W][16]
  long bstride;          // elements between samples
  const float* ss;       // [n][16 nchunks][2] (scale, shift) or nullptr
  int nchunks;
};

struct ClGeom {
  ClSrc src[2];
  int Cout, CoutP;            // Cout % 16 == 0 (CoutP == Cout)
  int D, H, W;
  int ntz, nty, ntx, ntiles;  // tiles per sample, ntiles = batch * ntz * nty * ntx
  int nchunks;                // src[0].nchunks + src[1].nchunks
  long y_bstride;             // elements between samples of y ([n][Cout / 16][D][H][W][16])
  float* stats;               // != nullptr: (count, mean, M2) summaries of y per (sample, channel, workgroup), layout of conv_bf16x3.h
  int nbatch;
};

// the value a consumer sees for a stored raw value v: InstanceNorm + ReLU folded into one fma and one max
__device__ __forceinline__ float cl_norm_relu(float v, float scale, float shift) { return fmaxf(fmaf(v, scale, shift), 0.f); }

// Per-lane running statistics of the workgroup's outputs: a lane holds channels 4 lq .. 4 lq + 3 of cout block j at position lr of its
// row blocks.  Shifted sums (shift = the first value the lane saw of that channel): n * var = s2 - s1^2 / n without the cancellation
// of raw sums; merged across lanes, waves and workgroups as (count, mean, M2) summaries (Chan et al.), always in the same order.
template <int NBT>
struct ClRunStats {
  float n, shift[NBT][4], s1[NBT][4], s2[NBT][4];
  __device__ __forceinline__ void reset() {
    n = 0.f;
#pragma unroll
    for (int j = 0; j < NBT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) shift[j][r] = s1[j][r] = s2[j][r] = 0.f;
  }
};

// End of the workgroup: every lane's (n, mean, M2) of its four channels, merged over the 16 positions lr by butterflies (symmetric:
// every lane of a 16-lane group ends with the same summary), then over the NW waves through LDS in wave order; written for every sample
// (zeros for the samples this workgroup did not compute).  All NW waves call it; `scratch`: NW x 16 NBT x 3 floats.
template <int NBT, int NW>
__device__ __forceinline__ void cl_stats_flush(const ClRunStats<NBT>& run, float* scratch, float* stats, int my_sample, int nbatch, int cout,
                                               int n0, int slots, int slot, int wid, int lane, int tid) {
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int j = 0; j < NBT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float n = run.n, mean = 0.f, m2 = 0.f;
      if (n > 0.f) {
        mean = run.shift[j][r] + run.s1[j][r] / n;
        m2 = run.s2[j][r] - run.s1[j][r] * run.s1[j][r] / n;
        if (m2 < 0.f) m2 = 0.f;
      }
#pragma unroll
      for (int sh = 1; sh <= 8; sh <<= 1) {
        const float nb = __shfl_xor(n, sh, 64), mb = __shfl_xor(mean, sh, 64), qb = __shfl_xor(m2, sh, 64);
        // symmetric merge: both partners compute the same sums in the same operand order (lower lane's summary first)
        const bool low = (lr & sh) == 0;
        float na = low ? n : nb, ma = low ? mean : mb, qa = low ? m2 : qb;
        const float nc = low ? nb : n, mc = low ? mb : mean, qc = low ? qb : m2;
        welford_merge(na, ma, qa, nc, mc, qc);
        n = na; mean = ma; m2 = qa;
      }
      if (lr == 0) {
        float* d = scratch + ((wid * NBT + j) * 16 + 4 * lq + r) * 3;
        d[0] = n; d[1] = mean; d[2] = m2;
      }
    }
  __syncthreads();
  if (tid < 16 * NBT) {
    const int j = tid >> 4, c = tid & 15, co = n0 + j * 16 + c;
    float n = 0.f, m = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* d = scratch + ((w * NBT + j) * 16 + c) * 3;
      welford_merge(n, m, q, d[0], d[1], d[2]);
    }
    if (co < cout)
      for (int b = 0; b < nbatch; ++b) {
        float* o = stats + (((long)b * cout + co) * slots + slot) * 3;
        const bool mine = b == my_sample;
        o[0] = mine ? n : 0.f; o[1] = mine ? m : 0.f; o[2] = mine ? q : 0.f;
      }
  }
}

// (count, mean, M2) summaries of `slots` workgroups per (sample, channel) -> mean, rstd and the (scale, shift) pair consumers apply on
// load.  One thread per row (rows = samples x channels, a few hundred at most), fixed merge order.
__global__ __launch_bounds__(64) void cl_stats_finalize_kernel(const float* __restrict__ stats, int rows, int slots, float eps,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                               float* __restrict__ ss) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float* s = stats + (long)row * slots * 3;
  float n = 0.f, m = 0.f, q = 0.f;
  for (int i = 0; i < slots; ++i) welford_merge(n, m, q, s[3 * i], s[3 * i + 1], s[3 * i + 2]);
  const float var = n > 0.f ? q / n : 0.f;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (mean_out) mean_out[row] = m;
  if (rstd_out) rstd_out[row] = rstd;
  ss[2 * row] = rstd;
  ss[2 * row + 1] = -m * rstd;
}

// NCDHW <-> CL16 (tests, probes and the boundaries of the converted levels).  C % 16 == 0.
__global__ __launch_bounds__(256) void cl16_from_ncdhw_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, long S,
                                                              long x_bstride, long y_bstride) {
  const long total = (long)N * C * S;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int c = (int)(it & 15);
    long r = it >> 4;
    const long pos = r % S;
    r /= S;
    const int cb = (int)(r % (C / 16)), n = (int)(r / (C / 16));
    y[(long)n * y_bstride + ((long)cb * S + pos) * 16 + c] = x[(long)n * x_bstride + (long)(cb * 16 + c) * S + pos];
  }
}
__global__ __launch_bounds__(256) void cl16_to_ncdhw_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, long S,
                                                            long x_bstride, long y_bstride) {
  const long total = (long)N * C * S;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long pos = it % S;
    long r = it / S;
    const int c = (int)(r % C), n = (int)(r / C);
    y[(long)n * y_bstride + (long)c * S + pos] = x[(long)n * x_bstride + ((long)(c >> 4) * S + pos) * 16 + (c & 15)];
  }
}

#if defined(CL_STAMPS)
// in-kernel stamps (probe builds): consumer waves 0 and 4 (one SIMD) and loader wave 8 of workgroup 0, work items 2..4
__device__ long long g_cl_stamps[3 * 3 * 16];
#define CL_STAMP(k)                                                                                                      \
  do {                                                                                                                   \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (wid == 0 || wid == 4 || wid == 8) && lane == 0 && item_no >= 2 && item_no < 5) \
      g_cl_stamps[((wid >> 2) * 3 + item_no - 2) * 16 + (k)] = clock64();                                                \
  } while (0)
#else
#define CL_STAMP(k) ((void)0)
#endif

// Eight consumer waves + four loader waves on 4 x 8 x 16 tiles (conv_bf16x3_ws.h), CL16 input and output.
template <int NBT>
__global__ __launch_bounds__(768) void conv3d_cl16_fwd_ws_kernel(const uint4* __restrict__ wsplit, const float* __restrict__ bias,
                                                                 float* __restrict__ y, ClGeom g) {
  typedef Bf3T<8> TC;
  constexpr int MB = 4;                                 // row blocks per consumer wave
  constexpr bool WHOLE = NBT == 1, PIPE_B2 = NBT < 3;
  constexpr int WPL = WHOLE ? 3 : 1;
  constexpr int NB = 16 * NBT, PX = TC::PX, PY = TC::PY, NPOSP = TC::NPOSP;
  constexpr int NC = 512, NL = 256;                     // consumer / loader threads
  constexpr int WITEMS = WPL * 6 * Bf3::SLOTS * NB, WU = (WITEMS + NC - 1) / NC;
  // loader staging: a group of 16 consecutive lanes takes 8 consecutive halo positions x 2 channel octets
  constexpr int GROUPS = (TC::NPOS + 7) / 8, ROUNDS = (GROUPS * 16 + NL - 1) / NL;
  ICL_DYN_LDS(uint4, lds);
  uint4* Xs = lds;
  uint4* Ws = lds + TC::XS_U4;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  int wid = tid >> 6;
  ICL_WAVE_UNIFORM(wid);
  const int n0 = blockIdx.y * NB;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const int tiles_per = g.ntz * g.nty * g.ntx;

  // zero the pad positions once (read by the zero slot: garbage * 0 must not be NaN)
  for (int i = tid; i < 6 * (NPOSP - TC::NPOS); i += NC + NL)
    Xs[(i / (NPOSP - TC::NPOS)) * NPOSP + TC::NPOS + (i % (NPOSP - TC::NPOS))] = make_uint4(0u, 0u, 0u, 0u);

#if defined(CL_DBG) && (CL_DBG & 8)
  if ((blockIdx.x >> 3) & 1) {      // odd workgroups of an XCD start half an item late
    for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  // tile order: every XCD walks its own contiguous eighth of the tile list (conv_bf16x3.h); gridDim.x % 8 == 0
  const int per_xcd = (g.ntiles + 7) / 8, xcd = blockIdx.x & 7, wgs_per_xcd = gridDim.x >> 3;
  const int xcd_end = (xcd + 1) * per_xcd < g.ntiles ? (xcd + 1) * per_xcd : g.ntiles;
  auto next_tile = [&](int t) { return t + wgs_per_xcd < xcd_end ? t + wgs_per_xcd : g.ntiles; };
  int tile = xcd * per_xcd + (blockIdx.x >> 3), chunk = 0;
  if (tile >= xcd_end) tile = g.ntiles;

  if (wid >= 8) {
    // ================================================================================================ loader waves
    const int lt = tid - NC;
    int item_no = -1;
    (void)item_no;
    ICL_SETPRIO(0);
    // Lane map inside a 16-lane group (positions p0 .. p0 + 7): lanes 0-3 positions 0-3 octet 0, lanes 4-7 positions 4-7 octet 1,
    // lanes 8-11 positions 4-7 octet 0, lanes 12-15 positions 0-3 octet 1.  The two 16-byte loads of the group cover its 512 contiguous
    // bytes; the ds_write_b128 of eight consecutive lanes (the hardware's conflict group, banks = address / 4 mod 32) land on eight
    // different 16-byte slots of a 128-byte bank row (the two octet images are a multiple of 256 bytes apart: 4 + 4 positions that
    // differ by 4 fill the row).
    const int l16 = lt & 15, sub = l16 >> 2, oct = sub & 1, hi4 = ((sub + 1) >> 1) & 1;
    // round r stages position p0 + 128 r (NL / 16 groups of 8 positions per round) into LDS slot oct * NPOSP + p0 + 128 r: the slot
    // offsets are immediates, and the halo coordinates of a round follow from the previous round's by carries (128 = 7 rows of 18 + 2),
    // so the loaders hold no per-round tables (the NCDHW kernel keeps 18 registers of them; here they would spill)
    static_assert(NL / 16 * 8 == 128 && PX == 18 && PY == 10, "coordinate increments of a staging round");
    const int p0 = 8 * (lt >> 4) + 4 * hi4 + (l16 & 3);
    const int px0 = p0 % PX, py0 = (p0 / PX) % PY, pz0 = p0 / (PX * PY);
    const bool last_ok = p0 + 128 * (ROUNDS - 1) < TC::NPOS;      // the last round covers the tile's tail
    uint4* const xdst = Xs + oct * NPOSP + p0;
    uint4 pl[ROUNDS][3];                                 // raw values (two quads), then the three packed planes of every staging item
    float sc[8], sh[8];                                  // (scale, shift) of the lane's channel octet in the chunk in flight
    icl_rsrc_t xr = icl_make_rsrc(g.src[0].p, 0u);
    int toff = 0, oz = 0, oy = 0, ox = 0;
    bool has_ss = false;
    unsigned okmask = 0u;
    auto origin = [&](int t, int ch) {
#if defined(CL_DBG) && (CL_DBG & 4)
      t = (int)(blockIdx.x & 7);
#endif
      const int b = t / tiles_per, bt = t % tiles_per;
      ox = (bt % g.ntx) * TC::TX; oy = ((bt / g.ntx) % g.nty) * TC::TY; oz = (bt / (g.ntx * g.nty)) * TC::TZ;
      int si = ch >= g.src[0].nchunks ? 1 : 0;
      ICL_WAVE_UNIFORM(si);
      const int cs = ch - (si ? g.src[0].nchunks : 0);
      const ClSrc& s = g.src[si];
      xr = icl_make_rsrc(s.p + (long)b * s.bstride + (long)cs * 16 * DHW, (unsigned)(16 * DHW * 4));
      // float offset of halo position (0, 0, 0) of the tile, octet included
      toff = (((oz - 1) * (int)HW + (oy - 1) * g.W + (ox - 1)) << 4) + oct * 8;
      has_ss = s.ss != nullptr;
      if (has_ss) {
        const float4* q = reinterpret_cast<const float4*>(s.ss + ((long)b * s.nchunks * 16 + cs * 16 + oct * 8) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = q[i];
          sc[2 * i] = v.x; sh[2 * i] = v.y; sc[2 * i + 1] = v.z; sh[2 * i + 1] = v.w;
        }
      }
    };
    auto issue_all = [&]() __attribute__((always_inline)) {
#if defined(CL_DBG) && (CL_DBG & 1)
      if (item_no >= 0) return;
#endif
      okmask = 0u;
      int px = px0, py = py0, pz = pz0;
      // (opaque: left visible, the compiler hoists the nine rounds' coordinates out of the tile loop as tables and spills them)
      ICL_OPAQUE_INT(px); ICL_OPAQUE_INT(py); ICL_OPAQUE_INT(pz);
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int gz = oz - 1 + pz, gy = oy - 1 + py, gx = ox - 1 + px;
        const bool ok = (r + 1 < ROUNDS || last_ok) & ((unsigned)gz < (unsigned)g.D) & ((unsigned)gy < (unsigned)g.H) & ((unsigned)gx < (unsigned)g.W);
        const unsigned boff = ok ? (unsigned)(toff + ((pz * (int)HW + py * g.W + px) << 4)) * 4u : 0x80000000u;
        okmask |= ok ? 1u << r : 0u;
        pl[r][0] = icl_buffer_load_u32x4(xr, boff);
        pl[r][1] = icl_buffer_load_u32x4(xr, boff, 16u);
        // position + 128
        px += 2;
        const int cx = px >= PX ? 1 : 0;
        px -= cx * PX;
        py += 7 + cx;
        const int cy = py >= PY ? 1 : 0;
        py -= cy * PY;
        pz += cy;
      }
    };
    auto split_range = [&](auto R0, auto R1) __attribute__((always_inline)) {
#pragma unroll
      for (int r = decltype(R0)::value; r < decltype(R1)::value; ++r)
        if (r < ROUNDS) {
#if defined(CL_DBG) && (CL_DBG & 2)
          ICL_PIN4(pl[r][0]); ICL_PIN4(pl[r][1]); continue;
#endif
          float v[8] = {__uint_as_float(pl[r][0].x), __uint_as_float(pl[r][0].y), __uint_as_float(pl[r][0].z), __uint_as_float(pl[r][0].w),
                        __uint_as_float(pl[r][1].x), __uint_as_float(pl[r][1].y), __uint_as_float(pl[r][1].z), __uint_as_float(pl[r][1].w)};
          if (has_ss) {
            // out-of-volume positions are padding of the NORMALISED tensor: 0, not relu(shift) — median(t, 0, u) with u = +inf inside
            // (= max(t, 0)) and u = 0 outside (= 0)
            const float u = (okmask >> r) & 1u ? __uint_as_float(0x7f800000u) : 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = icl_med3(fmaf(v[c], sc[c], sh[c]), 0.f, u);
          }
          bf3_split8(v, pl[r][0], pl[r][1], pl[r][2]);
        }
    };
    typedef std::integral_constant<int, 0> I0;
#if defined(CL_STAMPS)
    typedef std::integral_constant<int, 1> I1;          // probe builds: stamp 3 = the first round's data have arrived and are split
#else
    typedef std::integral_constant<int, ROUNDS / 3> I1;
#endif
    typedef std::integral_constant<int, ROUNDS> I2;
    auto deposit = [&]() {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        if (r + 1 == ROUNDS && !last_ok) continue;
        uint4* d = xdst + 128 * r;
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
    while (tile < g.ntiles) {
      int ntile = tile, nchunk = chunk + 1;
      if (nchunk == g.nchunks) { nchunk = 0; ntile = next_tile(tile); }
      const bool more = ntile < g.ntiles;
      ++item_no;
      CL_STAMP(0);
      __syncthreads();                         // (A) the LDS image of this item (and the consumers' weights) is complete
      CL_STAMP(1);
      if (more) { origin(ntile, nchunk); issue_all(); }
      CL_STAMP(2);
      if (!WHOLE) { __syncthreads(); __syncthreads(); }
      if (more) split_range(I0(), I1());
      if (!WHOLE) { __syncthreads(); __syncthreads(); }
      CL_STAMP(3);
      if (more) split_range(I1(), I2());
      CL_STAMP(4);
      __syncthreads();                         // (B) the consumers have finished reading this item's image
      CL_STAMP(5);
      if (more) deposit();
      CL_STAMP(6);
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
  const int lanepos = (wz * PY + wy) * PX + lr;
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
    for (int m = 0; m < MB; ++m) pa1[m] = xp[m * PX];
  };
  auto load_x23 = [&](int sdz, int spair) {
    const uint4* xp = frag_ptr(sdz, spair);
#pragma unroll
    for (int m = 0; m < MB; ++m) pa23[m][1] = xp[4 * NPOSP + m * PX];      // a3 first: its products lead the Y half
#pragma unroll
    for (int m = 0; m < MB; ++m) pa23[m][0] = xp[2 * NPOSP + m * PX];
  };

  if (tile < g.ntiles) {
    load_w(0, 0);
    if (WHOLE) store_w();
  }
  // bias of the lane's four output channels per cout block, loaded once (conv_bf16x3_ws.h)
  float4 bvs[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int co = n0 + j * 16 + 4 * lq;
    bvs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && co < g.Cout) bvs[j] = *reinterpret_cast<const float4*>(bias + co);
    ICL_PIN1(bvs[j].x); ICL_PIN1(bvs[j].y); ICL_PIN1(bvs[j].z); ICL_PIN1(bvs[j].w);
  }
  bool first_item = true;
  int item_no = -1;
  (void)item_no;
  ClRunStats<NBT> run;
  run.reset();
  int my_sample = -1;
  while (tile < g.ntiles) {
    int ntile = tile, nchunk = chunk + 1;
    if (nchunk == g.nchunks) { nchunk = 0; ntile = next_tile(tile); }
    ++item_no;
    CL_STAMP(0);
    if (WHOLE && g.nchunks > 1) {
      // between the barriers B of the last item and A of this one (see conv_bf16x3_ws.h)
      if (!first_item) store_w();
      if (ntile < g.ntiles) load_w(nchunk, 0);
    }
    first_item = false;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      if (WHOLE) {
        if (dz == 0) {
          __syncthreads();                     // (A)
          CL_STAMP(1);
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
        // operands swapped against conv_bf16x3_ws.h: A = weight fragment (rows = output channels), B = activation fragment (columns =
        // positions) — the same products summed over the same k in the same order, the accumulator transposed
#pragma unroll
        for (int sb = 2; sb >= 0; --sb)
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int j = 0; j < NBT; ++j) acc[m][j] = icl_mfma_16x16x32_bf16(pb[cur][sb][j], pa1[m], acc[m][j]);
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
            for (int j = 0; j < NBT; ++j) acc[m][j] = icl_mfma_16x16x32_bf16(pb[cur][sbb[t]][j], pa23[m][sa[t]], acc[m][j]);
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
      CL_STAMP(2 + dz);
    }
    // ---- epilogue (placement as in conv_bf16x3_ws.h: the early wave of a SIMD in front of barrier B, the late one behind it).
    // Lane: position lr of row block m, channels 4 lq .. 4 lq + 3 of cout block j: one 16-byte store per (m, j).
    auto epilogue = [&]() __attribute__((always_inline)) {
      const int b = tile / tiles_per, bt = tile % tiles_per;
      const int x0 = (bt % g.ntx) * TC::TX, y0 = ((bt / g.ntx) % g.nty) * TC::TY, z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
      const int gz = z0 + wz, gx = x0 + lr;
      float* yb = y + (long)b * g.y_bstride + ((long)gz * HW + gx) * 16 + 4 * lq;
      float4 bv[NBT];
      float* yc[NBT];
      bool cok[NBT];
#pragma unroll
      for (int j = 0; j < NBT; ++j) {
        const int co = n0 + j * 16 + 4 * lq;
        cok[j] = co < g.Cout;
        bv[j] = bvs[j];
        yc[j] = yb + (long)(co >> 4) * DHW * 16;
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const bool okm = gz < g.D && y0 + wy + m < g.H && gx < g.W;
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
          const float4 v = make_float4(acc[m][j][0] + bv[j].x, acc[m][j][1] + bv[j].y, acc[m][j][2] + bv[j].z, acc[m][j][3] + bv[j].w);
          if (okm && cok[j]) {
            *reinterpret_cast<float4*>(yc[j] + (long)(y0 + wy + m) * g.W * 16) = v;
            if (g.stats) {
              const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (run.n == 0.f) run.shift[j][r] = vv[r];      // the first value this lane sees of the channel
                const float d = vv[r] - run.shift[j][r];
                run.s1[j][r] += d;
                run.s2[j][r] = fmaf(d, d, run.s2[j][r]);
              }
            }
          }
          acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (okm) run.n += 1.f;      // (counts positions: the same for every channel the lane holds)
      }
      my_sample = b;
    };
    const bool last_chunk = chunk == g.nchunks - 1;
    if (last_chunk && wid < 4) epilogue();
    CL_STAMP(5);
    __syncthreads();                           // (B)
    CL_STAMP(6);
    if (last_chunk && wid >= 4) epilogue();
    tile = ntile;
    chunk = nchunk;
  }
  // (the loader waves have returned: the barrier inside counts the eight consumer waves)
  if (g.stats)
    cl_stats_flush<NBT, 8>(run, reinterpret_cast<float*>(Ws + WPL * TC::ws_u4(NB)), g.stats, my_sample, g.nbatch, g.Cout, n0, (int)gridDim.x,
                           (int)blockIdx.x, wid, lane, tid);
}

}  // namespace icl
