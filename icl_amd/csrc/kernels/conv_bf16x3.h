// conv_bf16x3.h — 3x3x3 convolution (forward / input gradient) with fp32 operands split into three bf16 terms.
//
// The fp32 matrix pipe of gfx950 peaks at 157 TFLOP/s and the 3x3x3 convolutions of the U-Net are bound by it (profiles/r2_pmc_conv.md:
// 81-85 % busy).  The bf16 pipe is 16x faster.  Every fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significand
// bits, round-to-nearest split x = x1 + x2 + x3, bf3_split2 below), so a product x*w is the sum of nine bf16 products, each exact
// in fp32; the kernel accumulates the six largest,
//     x1 w1 + (x1 w2 + x2 w1) + (x1 w3 + x2 w2 + x3 w1),
// and drops x2 w3 + x3 w2 + x3 w3, bounded by 2^-23 |x w| (|x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|).  Measured on 4 M random pairs
// (tools/split_error.py): error of the six-term product relative to |x w|, in units of 2^-24 (an fp32 multiply: mean 0.36, max
// 1.0): round-to-nearest splits mean 0.06, max 0.9, mean SIGNED error 0.000 — the same on N(0,1), on all-positive and on
// log-uniform 1e-15..1e15 operands; the truncation splits of round 2 measured mean 0.69, max 7.1 with every error on the sign of
// -x w (a bias that adds up coherently over post-ReLU activations times same-sign weights).
// Accumulation is fp32 inside v_mfma_f32_16x16x32_bf16, as in the fp32 kernels.  Six bf16 MFMAs of 16 cycles replace eight fp32 MFMAs
// of 32 cycles for the same 16 x 16 x 32 block of multiply-adds: 2.67x the matrix rate at fp32 accuracy.
//
// GEMM view (same as conv_mfma_static.h): rows = 16 consecutive x positions, columns = 16 output channels, k = (tap, input channel).
// One MFMA takes k = 32: two taps x 16 input channels; lane group lq = lane >> 4 supplies 8 consecutive k: tap (lq >> 1) of the
// pair, channels 8 (lq & 1) .. + 7.  Both operands are therefore kept "8 channels contiguous" (16 bytes of bf16) in LDS and every
// operand fragment is ONE ds_read_b128:
//   Xs[split 3][half 2][position][8 ch]   halo tile of (TZ+2) x (TY+2) x 18 positions of a 16-channel chunk
//   Ws[split 3][half 2][slot 10][cout][8 ch]   ten consecutive taps (one of three stages: 27 taps + a zero slot = 14 pairs) for the cout block
// The fp32 NCDHW input is split and transposed while it is staged (a thread reads one position of 8 channel planes, 3 ds_write_b128);
// the weights come from the fp32 pack of the fp32 kernels (PackedWeights), split by a small kernel in front of the convolution.  A workgroup (8 waves) owns output tiles of 4 x 8 x 16 voxels and
// walks a list of tiles; the global loads of the next (tile, channel chunk) are in flight while the current one is multiplied.
// Reference op: nn.Conv3d(k=3, pad=1) inside UnetConv3 (/root/reference/code/networks/utils.py:104,107) and its input gradient.
#pragma once

namespace icl {

struct Bf3Geom {
  int Cin, Cout, CinP, CoutP; // Cin % 16 == 0; CinP / CoutP: extents of the packed weights wp[27 taps][CinP][CoutP] (conv_mfma.h)
  int D, H, W;                // W % 4 == 0
  int ntz, nty, ntx, ntiles;  // tiles per sample, ntiles = batch * ntz * nty * ntx
  int nchunks;                // Cin / 16
  long x_bstride, y_bstride;
  float* stats;               // != nullptr: InstanceNorm statistics of the output, produced in the epilogue (bf3_stats_* below)
  int nbatch;                 // samples (the statistics of a workgroup are written for every sample: zeros for the ones it never sees)
  int wgs;                    // workgroups along x = statistics slots per (sample, channel): computed ONCE by the launcher (grid.x)
  int flags;                  // bit 1: non-temporal output stores (round 4, batch 2: 16->48 @96^3 348 vs 368 us, 32->32 @48^3 67.7 vs
                              // 69.7, 16->16 @96^3 137 vs 139: the launcher sets it for three cout blocks).  (Bit 0 was a y-slowest
                              // tile order inside an XCD's share of the tile list, so that z- and x-neighbours run at the same time
                              // and their shared halo lines hit the XCD's L2: measured +-1 % on every layer, removed —
                              // profiles/r4_tile_order_nt_probe.txt.)
                              // bit 2 (round 6): plain tile order — workgroup blockIdx.x owns tile blockIdx.x and nothing else (grid.x =
                              // ntiles): the split-K launches of the deep levels, whose 18-72 tiles do not divide into eight XCD shares
  int ksplit;                 // > 1: blockIdx.z owns the channel chunks [z nchunks / ksplit, (z + 1) nchunks / ksplit) and writes its raw
  float* slab;                // partial sums (no bias, no statistics) to slab + z * slab_stride as a dense [n][Cout][D][H][W] tensor;
  long slab_stride;           // splitk_reduce_kernel (conv_mfma.h) adds the slices in a fixed order (+ bias) into y
};

// Output tile 4 x TY x 16 voxels, one wave per four (z, y) rows: TY = 8 -> 8 waves, 120-150 KB of LDS (one workgroup per CU);
// TY = 4 -> 4 waves, 78 KB with one cout block: two workgroups per CU whose staging and multiply phases interleave.
struct Bf3Base { static constexpr int SLOTS = 10; };
template <int TY_>
struct Bf3T : Bf3Base {
  static constexpr int TZ = 4, TY = TY_, TX = 16, PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
  static constexpr int NPOS = PZ * PY * PX;
  static constexpr int NPOSP = (NPOS + 1 + 15) / 16 * 16;        // the zero slot of the last pair reads one position past the tile;
                                                                 // a multiple of 16 keeps the two channel halves 256 B apart (banks)
  static constexpr int NW = TZ * TY / 4, NT = 64 * NW, ITEMS = 2 * NPOS, ROUNDS = (ITEMS + NT - 1) / NT;
  static constexpr int MB = 4;                                   // row blocks (16 x positions) per wave
  static constexpr int XS_U4 = 6 * NPOSP;                        // uint4 (8 bf16) units
  static constexpr int ws_u4(int nb) { return 6 * SLOTS * nb; }
  static constexpr size_t lds_bytes(int nbt, int planes = 1) { return (size_t)(XS_U4 + planes * ws_u4(16 * nbt)) * 16 + 8 * 48 * 3 * 4; }
};
// Flat tile for rows of 24 voxels (the 24^3 level, round 3): 2 x 8 x 24 = 384 outputs = 24 row blocks of 16 consecutive positions of
// the flattened (z, y, x) index — three per wave — instead of 4 x 8 x 16 tiles whose second x tile is half empty.  A row block may wrap
// from one row to the next (24 = 16 + 8): a lane's halo offsets of its three blocks are per-lane constants, not multiples of the pitch.
struct Bf3F24 : Bf3Base {
  static constexpr int TZ = 2, TY = 8, TX = 24, PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
  static constexpr int NPOS = PZ * PY * PX;
  static constexpr int NPOSP = (NPOS + 1 + 15) / 16 * 16;
  static constexpr int NW = 8, NT = 64 * NW, ITEMS = 2 * NPOS, ROUNDS = (ITEMS + NT - 1) / NT;
  static constexpr int MB = 3;
  static constexpr int XS_U4 = 6 * NPOSP;
  static constexpr int ws_u4(int nb) { return 6 * SLOTS * nb; }
  static constexpr size_t lds_bytes(int nbt, int planes = 1) { return (size_t)(XS_U4 + planes * ws_u4(16 * nbt)) * 16 + 8 * 48 * 3 * 4; }
};
// Flat tile for rows of 12 voxels (the 12^3 level, round 6): 4 x 4 x 12 = 192 outputs = 12 row blocks, three per wave of a FOUR-wave
// workgroup.  78 KB of LDS with two cout blocks (no statistics scratch: these launches are split over the channel chunks and hand raw
// partial sums to splitk_reduce_kernel) and <= 256 registers: two workgroups per CU, one staging while the other multiplies.
struct Bf3F12 : Bf3Base {
  static constexpr int TZ = 4, TY = 4, TX = 12, PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
  static constexpr int NPOS = PZ * PY * PX;
  static constexpr int NPOSP = (NPOS + 1 + 15) / 16 * 16;
  static constexpr int NW = 4, NT = 64 * NW, ITEMS = 2 * NPOS, ROUNDS = (ITEMS + NT - 1) / NT;
  static constexpr int MB = 3;
  static constexpr int XS_U4 = 6 * NPOSP;
  static constexpr int ws_u4(int nb) { return 6 * SLOTS * nb; }
  static constexpr size_t lds_bytes(int nbt, int planes = 1) { return (size_t)(XS_U4 + planes * ws_u4(16 * nbt)) * 16; }
};
// Flat tile for a WHOLE 6^3 volume (the centre of the U-Net, round 6): 216 outputs on 16 row blocks of the flattened index — two per wave
// of an eight-wave workgroup (four per wave on four waves spilled 140 B per lane), the last 40 rows idle (their operand reads run past
// the halo image into the weight planes: finite garbage in matrix rows that are never stored).  The flattened index of the tile IS the element offset inside the volume, so an epilogue quad
// that wraps from one row of 6 to the next is still one contiguous 16-byte store.  Split launches only, as Bf3F12.
struct Bf3F6 : Bf3Base {
  static constexpr int TZ = 6, TY = 6, TX = 6, PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
  static constexpr int NPOS = PZ * PY * PX;
  static constexpr int NPOSP = (NPOS + 1 + 15) / 16 * 16;
  static constexpr int NW = 8, NT = 64 * NW, ITEMS = 2 * NPOS, ROUNDS = (ITEMS + NT - 1) / NT;
  static constexpr int MB = 2;
  static constexpr int XS_U4 = 6 * NPOSP;
  static constexpr int ws_u4(int nb) { return 6 * SLOTS * nb; }
  static constexpr size_t lds_bytes(int nbt, int planes = 1) { return (size_t)(XS_U4 + planes * ws_u4(16 * nbt)) * 16; }
};
// waves of the workgroup for the template arguments (TY, FLAT) of conv3d_bf16x3_fwd_kernel: TY = 6 names the 6^3 tile (eight waves)
constexpr int bf3_waves(int ty, bool flat) { return flat && ty == 6 ? 8 : ty; }
typedef Bf3Base Bf3;

// Two fp32 -> their three bf16 terms, packed pairwise (low half = a).  Round-to-nearest splits: s1 = rn(v), s2 = rn(v - s1),
// s3 = v - s1 - s2.  Both subtractions are exact (v - s1 has at most 16 significant bits, v - s1 - s2 at most 8), so
// v = s1 + s2 + s3 EXACTLY for every finite v below the bf16 overflow threshold, with |s2| <= 2^-8 |v|, |s3| <= 2^-16 |v| and
// residual signs that vary from element to element (a truncation split keeps all three terms on the sign of v: the dropped
// product terms then add up coherently over a sum of same-sign products).  11 VALU per pair (3 v_cvt_pk_bf16_f32, 4 unpack, 4 sub).
// Non-finite v (and |v| > 3.39e38, which rounds to a bf16 infinity): s2 and s3 are NaN — the result is non-finite wherever the
// fp32 kernels give a non-finite result, as NaN where they may give an infinity.
// (Round 6: hipcc's SLP vectoriser packs the two residual subtractions of a pair into one `v_pk_add_f32 ... neg_lo neg_hi`.  That is the
// faster form wherever the split runs while the wave's own SIMD multiplies nothing — every kernel of this file and the weight-gradient
// kernels: 0-5 % slower without it — and the slower one in the loader waves of conv3d_bf16x3_fwd_ws_kernel, which is therefore compiled
// in a unit of its own with the vectoriser off: csrc/icl_hip_noslp.hip, profiles/r6_pk_add_ab.txt.)
__device__ __forceinline__ void bf3_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = icl_pack_bf16_rn(a, b);
  const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
  p2 = icl_pack_bf16_rn(ra, rb);
  const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u);
  p3 = icl_pack_bf16_rn(sa, sb);
}
// 8 values -> 8 packed bf16 per split
__device__ __forceinline__ void bf3_split8(const float* v, uint4& o1, uint4& o2, uint4& o3) {
  bf3_split2(v[0], v[1], o1.x, o2.x, o3.x);
  bf3_split2(v[2], v[3], o1.y, o2.y, o3.y);
  bf3_split2(v[4], v[5], o1.z, o2.z, o3.z);
  bf3_split2(v[6], v[7], o1.w, o2.w, o3.w);
}

// ---- InstanceNorm statistics of the convolution output, from the epilogue's registers (round 4; reference: Conv3d -> InstanceNorm3d,
// /root/reference/code/networks/utils.py:104-105).  The stand-alone statistics pass (norm.h rowstats_partial_kernel) re-reads the whole
// output (113 MB for a 16-channel 96^3 x 2 tensor) to produce per-(sample, channel) chunk summaries (count, mean, M2) that the
// normalisation kernel merges; here every WORKGROUP produces one such summary per output channel over all the tiles it computed — from
// the fp32 values it is about to store — and the normalisation kernel merges gridDim.x of them: the same (count, mean, M2) format, the
// same merge (Chan et al.), fixed order, no atomics.  Requires that a workgroup's tile list stays inside one sample (launcher).
// Layout: stats[((sample * Cout + co) * slots + blockIdx.x) * 3 + {0, 1, 2}], slots = gridDim.x.
template <int NBT>
struct Bf3RunStats {
  float n[NBT], mean[NBT], m2[NBT];      // of the tiles so far, per cout block j: channel n0 + 16 j + (lane & 15); equal in the four lane groups
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int j = 0; j < NBT; ++j) n[j] = mean[j] = m2[j] = 0.f;
  }
};
// one tile's values of cout block j in this lane: v[0 .. cnt), the first `cnt` of them valid (cnt is a multiple of 4 or 0 per row block)
template <int NBT>
__device__ __forceinline__ void bf3_stats_add(Bf3RunStats<NBT>& run, int j, const float* v, const bool* ok, int nv) {
  float n = 0.f, shift = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < nv && ok[i >> 2]) {
      if (n == 0.f) shift = v[i];
      const float d = v[i] - shift;
      s1 += d;
      s2 += d * d;
      n += 1.f;
    }
  }
  float mean = 0.f, m2 = 0.f;
  if (n > 0.f) {
    mean = shift + s1 / n;
    m2 = s2 - s1 * s1 / n;
    if (m2 < 0.f) m2 = 0.f;
  }
  // the four lane groups (lane >> 4) hold four x quads of the same channel: symmetric merges, every lane ends with the same value
#pragma unroll
  for (int sh = 16; sh <= 32; sh <<= 1) {
    const float nb = __shfl_xor(n, sh, 64), mb = __shfl_xor(mean, sh, 64), qb = __shfl_xor(m2, sh, 64);
    float na = n, ma = mean, qa = m2;
    welford_merge(na, ma, qa, nb, mb, qb);
    if (na > 0.f) { n = na; mean = ma; m2 = qa; }
  }
  welford_merge(run.n[j], run.mean[j], run.m2[j], n, mean, m2);
}
// End of the workgroup: the NW waves' summaries through LDS (`scratch`: NW x 16 NBT x 3 floats), merged in wave order by the first 16 NBT
// threads, written for every sample (zeros for the samples this workgroup did not compute).  All (live) waves of the workgroup call it.
template <int NBT, int NW>
__device__ __forceinline__ void bf3_stats_flush(const Bf3RunStats<NBT>& run, float* scratch, float* stats, int my_sample, int nbatch, int cout,
                                                int n0, int slots, int slot, int wid, int lane, int tid) {
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < NBT; ++j) {
      float* d = scratch + ((wid * NBT + j) * 16 + lane) * 3;
      d[0] = run.n[j]; d[1] = run.mean[j]; d[2] = run.m2[j];
    }
  }
  __syncthreads();
  if (tid < 16 * NBT) {
    const int j = tid >> 4, lr = tid & 15, co = n0 + j * 16 + lr;
    float n = 0.f, m = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* d = scratch + ((w * NBT + j) * 16 + lr) * 3;
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
constexpr size_t kBf3StatsLdsBytes = 8 * 48 * 3 * 4;      // scratch of bf3_stats_flush behind the weight planes (every launch reserves it)

#ifndef ICL_SECOND_UNIT      // (the two non-template kernels of this file are defined by the main unit only: csrc/icl_hip_noslp.hip)
// Split weights, ready for LDS: ws[chunk][stage 3][split * 2 + half][slot 10][CoutP] of 8 packed bf16, from the fp32 pack
// wp[tap][CinP][CoutP] of the fp32 kernels: slot s of stage g is tap 10 g + s; the 28th slot (stage 2, slot 7) is the zero partner of
// tap 26, slots 8 / 9 of stage 2 are unused.  One small launch in front of
// the convolution (the matrices are 7 KB .. 1.3 MB): splitting them inside the convolution, once per tile and dz plane, cost as
// many VALU instructions as splitting the activations.
__global__ __launch_bounds__(256) void conv_bf16x3_split_weights_kernel(const float* __restrict__ wp, uint4* __restrict__ ws, int cinP, int coutP,
                                                                        int nchunks) {
  const long total = (long)nchunks * 3 * 2 * Bf3::SLOTS * coutP;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int co = (int)(it % coutP);
    long r = it / coutP;
    const int slot = (int)(r % Bf3::SLOTS);
    r /= Bf3::SLOTS;
    const int hf = (int)(r % 2);
    r /= 2;
    const int dz = (int)(r % 3), chunk = (int)(r / 3);
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 10 * dz + slot < 27 ? wp[((long)(10 * dz + slot) * cinP + chunk * 16 + hf * 8 + c) * coutP + co] : 0.f;
    uint4 o1, o2, o3;
    bf3_split8(v, o1, o2, o3);
    const long plane = (long)Bf3::SLOTS * coutP;
    uint4* d = ws + ((long)(chunk * 3 + dz) * 6 + hf) * plane + (long)slot * coutP + co;
    d[0] = o1;
    d[2 * plane] = o2;
    d[4 * plane] = o3;
  }
}

// Schedule of conv3d_bf16x3_fwd_kernel (the template parameter V is kept at 60 = 8 + 16 + 32 + 4 for the probes that name it; the other
// variants were removed in round 5 — measurements in DESIGN.md §4 and profiles/r3_pmc_conv.md):
//   8   (one cout block) all three weight planes of the channel chunk stay in LDS (43 KB beside the 104 KB halo tile): two barriers per
//       work item instead of six (round 3, batch 2: 16->16 @96^3 137 vs 152 us, 48->16 406 vs 423);
//   16  operand fragments fetched one half-step ahead; 32: every fragment read issued behind one MFMA of the half before it
//       (scheduling groups); 4: the halo loads of the next work item as buffer loads (hardware zero fill, no branches), one staging round
//       woven into each tap pair of the dz = 0 stage.  Together -7..13 % against 8 alone.
//   Measured and removed: the operand split pinned into the multiply phase (V & 2: +6-8 % time), straight-line loads with selects
//   (V & 1: +7 %), one weight plane per dz stage for one cout block (V = 0), the un-pipelined pair loop (V < 16).
// The same for up to kSplitMulti packed weights in ONE launch (grid.y = weight): a training step splits every convolution weight
// once, up front (they only change in the optimiser), instead of one small launch in front of every convolution call.
constexpr int kSplitMulti = 32;
struct SplitMulti {
  const float* wp[kSplitMulti];
  uint4* ws[kSplitMulti];
  int cinP[kSplitMulti], coutP[kSplitMulti], nchunks[kSplitMulti];
};
__global__ __launch_bounds__(256) void conv_bf16x3_split_weights_multi_kernel(SplitMulti m) {
  const int e = blockIdx.y;
  const float* __restrict__ wp = m.wp[e];
  uint4* __restrict__ ws = m.ws[e];
  const int cinP = m.cinP[e], coutP = m.coutP[e];
  const long total = (long)m.nchunks[e] * 3 * 2 * Bf3::SLOTS * coutP;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int co = (int)(it % coutP);
    long r = it / coutP;
    const int slot = (int)(r % Bf3::SLOTS);
    r /= Bf3::SLOTS;
    const int hf = (int)(r % 2);
    r /= 2;
    const int dz = (int)(r % 3), chunk = (int)(r / 3);
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 10 * dz + slot < 27 ? wp[((long)(10 * dz + slot) * cinP + chunk * 16 + hf * 8 + c) * coutP + co] : 0.f;
    uint4 o1, o2, o3;
    bf3_split8(v, o1, o2, o3);
    const long plane = (long)Bf3::SLOTS * coutP;
    uint4* d = ws + ((long)(chunk * 3 + dz) * 6 + hf) * plane + (long)slot * coutP + co;
    d[0] = o1;
    d[2 * plane] = o2;
    d[4 * plane] = o3;
  }
}
#endif  // ICL_SECOND_UNIT

#if defined(BF3_DEBUG) && (BF3_DEBUG & 16)
// in-kernel stamps (debug builds only): waves 0 and 4 of workgroup 0 (they share a SIMD) record s_memtime at the phase boundaries of
// work items 2..4; read back with icl_debug_bf3_stamps (tools/bf3_stamps.py)
__device__ long long g_bf3_stamps[2 * 3 * 32];
#define BF3_STAMP(k)                                                                                          \
  do {                                                                                                        \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (wid & 3) == 0 && lane == 0 && item_no >= 2 && item_no < 5)    \
      g_bf3_stamps[((wid >> 2) * 3 + item_no - 2) * 32 + (k)] = clock64();                                    \
  } while (0)
#else
#define BF3_STAMP(k) ((void)0)
#endif


template <int NBT, int TY, int V = 60, bool FLAT = false>
__global__ __launch_bounds__(64 * bf3_waves(TY, FLAT), 2) void conv3d_bf16x3_fwd_kernel(const float* __restrict__ x, const uint4* __restrict__ wsplit,
                                                                const float* __restrict__ bias, float* __restrict__ y, Bf3Geom g) {
  typedef typename std::conditional<FLAT, typename std::conditional<TY == 8, Bf3F24, typename std::conditional<TY == 4, Bf3F12, Bf3F6>::type>::type,
                                    Bf3T<TY>>::type TC;
  constexpr int MB = TC::MB;                            // row blocks per wave
  static_assert(V == 60, "one schedule (see the comment above the weight-split kernels)");
  static_assert(!FLAT || TY == 8 || TY == 4 || TY == 6, "flat tiles: 2 x 8 x 24 and 6 x 6 x 6 on eight waves, 4 x 4 x 12 on four");
  static_assert(TC::NW == bf3_waves(TY, FLAT), "one wave per four (z, y) rows / MB flat row blocks");
  // one cout block: all three weight planes of a chunk resident in LDS; B fragments double-buffered up to two cout blocks (with three
  // the second set does not fit the register budget: B of the next pair is then read behind the last MFMA)
  constexpr bool WHOLE = NBT == 1, PIPE_B2 = NBT < 3;
  static_assert(TC::ROUNDS <= 5, "one staging round per tap pair of the first dz stage");
  constexpr int WPL = WHOLE ? 3 : 1;                    // weight planes staged together
  constexpr int NB = 16 * NBT, PX = TC::PX, PY = TC::PY, NPOSP = TC::NPOSP, ROUNDS = TC::ROUNDS, NT = TC::NT;
  constexpr int WITEMS = WPL * 6 * Bf3::SLOTS * NB, WU = (WITEMS + NT - 1) / NT;      // 16-byte weight slots per staging step
  ICL_DYN_LDS(uint4, lds);
  uint4* Xs = lds;
  uint4* Ws = lds + TC::XS_U4;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int half = lq & 1, tp = lq >> 1;
  const int n0 = blockIdx.y * NB;
  const long HW = (long)g.H * g.W, DHW = g.D * HW;
  const int tiles_per = g.ntz * g.nty * g.ntx;

  // zero the pad positions once (read by the zero slot: garbage * 0 must not be NaN)
  for (int i = tid; i < 6 * (NPOSP - TC::NPOS); i += NT)
    Xs[(i / (NPOSP - TC::NPOS)) * NPOSP + TC::NPOS + (i % (NPOSP - TC::NPOS))] = make_uint4(0u, 0u, 0u, 0u);

  // ---- staging: item = (channel octet, halo position); lanes walk the positions of the tile in LDS order, so the three 16-byte
  // writes of an item land on consecutive LDS slots across the lanes (no bank conflicts) and the 8 loads of a wave-instruction
  // read runs of 18 consecutive floats.  Tile-invariant part of the addressing:
  int s_zyx[ROUNDS], s_dst[ROUNDS], s_rel[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int it = tid + r * NT;
    const int o = it / TC::NPOS, pos = it % TC::NPOS;
    const int px = pos % PX, row = pos / PX, py = row % PY, pz = row / PY;
    s_zyx[r] = it < TC::ITEMS ? (pz << 16) | (py << 8) | px : -1;
    s_dst[r] = o * NPOSP + pos;
    s_rel[r] = o * 8 * (int)DHW + (pz - 1) * (int)HW + (py - 1) * g.W + (px - 1);
  }
  float xv[ROUNDS][8];
  // the (sample, z0, y0, x0) of a tile: five integer divisions of wave-uniform values (a reciprocal sequence on the VALU each) — once
  // per work item, not once per staging round
  struct Origin { const float* xb; int z0, y0, x0, b, chunk; };
  auto origin_of = [&](int tile, int chunk) {
    const int b = tile / tiles_per, bt = tile % tiles_per;
    Origin o;
    o.x0 = (bt % g.ntx) * TC::TX, o.y0 = ((bt / g.ntx) % g.nty) * TC::TY, o.z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
    o.xb = x + (long)b * g.x_bstride + (long)chunk * 16 * DHW;
    o.b = b; o.chunk = chunk;
    return o;
  };
  auto load_x_at = [&](const Origin& o, int r0, int r1) {
    const float* xb = o.xb;
    const int z0 = o.z0, y0 = o.y0, x0 = o.x0;
#pragma unroll
    for (int r = r0; r < r1; ++r) {
      const int gz = z0 - 1 + (s_zyx[r] >> 16), gy = y0 - 1 + ((s_zyx[r] >> 8) & 255), gx = x0 - 1 + (s_zyx[r] & 255);
      // (bitwise &: the short-circuit form compiles to a chain of exec-mask branches)
      const bool ok = (s_zyx[r] >= 0) & ((unsigned)gz < (unsigned)g.D) & ((unsigned)gy < (unsigned)g.H) & ((unsigned)gx < (unsigned)g.W);
      // buffer loads: 32-bit lane offset against a descriptor of the 16-channel chunk (16 channels x D*H*W < 2^31 elements, checked by
      // the launcher); out-of-volume and idle lanes are handed offset 2^31 and the hardware returns 0 — no branches, no clamps.  Lane
      // part: the tile-invariant offset of the lane's halo position and channel octet (s_rel) + the tile's origin; the channel plane c
      // enters as the scalar offset of the load
      const icl_rsrc_t xr = icl_make_rsrc(xb, (unsigned)(16 * DHW * 4));
      const unsigned boff = ok ? (unsigned)(s_rel[r] + (z0 * (int)HW + y0 * g.W + x0)) * 4u : 0x80000000u;
#pragma unroll
#if defined(BF3_DEBUG) && (BF3_DEBUG & 32)
      for (int c = 0; c < 8; ++c) xv[r][c] = __uint_as_float(boff + c);      // ablation: the address arithmetic without the loads
#else
      for (int c = 0; c < 8; ++c) xv[r][c] = icl_buffer_load_f32(xr, boff, (unsigned)c * (unsigned)DHW * 4u);
#endif
    }
  };
  auto load_x = [&](int tile, int chunk) { load_x_at(origin_of(tile, chunk), 0, ROUNDS); };
  auto store_x = [&]() {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      if (s_zyx[r] < 0) continue;
      uint4* d = Xs + s_dst[r];
      uint4 o1, o2, o3;
      bf3_split8(xv[r], o1, o2, o3);
      d[0] = o1;
      d[2 * NPOSP] = o2;
      d[4 * NPOSP] = o3;
    }
  };
  // weights: already split (conv_bf16x3_split_weights_kernel): one dz plane = 60 NB slots of 16 bytes, copied through registers
  uint4 wv[WU];
  auto load_w = [&](int chunk, int dz) {      // WHOLE: dz = 0 and all three planes (they are contiguous in the workspace)
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

  // ---- operand bases: wave w owns the (z, y) rows 4 w .. 4 w + 3 of the tile
  const int wz = (4 * wid) / TY, wy = (4 * wid) % TY;
  // halo offset of the lane's position in row block m of the wave, relative to block 0 (FLAT: block 3 wid + m of the flattened tile)
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
    for (int m = 0; m < MB; ++m) moff[m] = m * PX;      // compile-time constants after unrolling
  }
  const uint4* xa = Xs + half * NPOSP + lanepos;        // + split * 2 * NPOSP + tap offset
  const uint4* wb = Ws + (half * Bf3::SLOTS + tp) * NB + lr;

  f32x4 acc[4][NBT];
  uint4 pa1[4], pa23[4][2], pb[PIPE_B2 ? 2 : 1][3][NBT];      // fragments fetched one half-step ahead
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < NBT; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Tile order: workgroup b runs on XCD b % 8 (its own L2).  Every XCD walks its own contiguous eighth of the tile list, the
  // workgroups of an XCD side by side in it, so that tiles which share halo planes are staged through the same L2 at about the
  // same time (FETCH_SIZE: the halo re-reads otherwise all go to HBM / MALL).  Needs gridDim.x % 8 == 0 (launcher).
  // (flags bit 2: one tile per workgroup, tile = blockIdx.x — the split-K launches of the deep levels)
  const bool plain = (g.flags & 4) != 0;
  const int per_xcd = (g.ntiles + 7) / 8, xcd = blockIdx.x & 7, wgs_per_xcd = gridDim.x >> 3;
  const int xcd_end = plain ? 0 : ((xcd + 1) * per_xcd < g.ntiles ? (xcd + 1) * per_xcd : g.ntiles);
  auto next_tile = [&](int t) { return t + wgs_per_xcd < xcd_end ? t + wgs_per_xcd : g.ntiles; };
  // split over the channel chunks: this workgroup's slice [c_lo, c_hi) and where its partial sums go
  const bool ksp = g.ksplit > 1;
  const int c_lo = ksp ? (int)((long)blockIdx.z * g.nchunks / g.ksplit) : 0;
  const int c_hi = ksp ? (int)((long)(blockIdx.z + 1) * g.nchunks / g.ksplit) : g.nchunks;
  const int c_n = c_hi - c_lo;
  float* const yout = ksp ? g.slab + (long)blockIdx.z * g.slab_stride : y;
  const long yout_bstride = ksp ? (long)g.Cout * DHW : g.y_bstride;
  int tile = plain ? (int)blockIdx.x : xcd * per_xcd + (blockIdx.x >> 3), chunk = c_lo;
  if (!plain && tile >= xcd_end) tile = g.ntiles;
  if (c_n <= 0) tile = g.ntiles;
  if (tile < g.ntiles) {
    load_w(c_lo, 0);
    load_x(tile, c_lo);
  }
  // the bias of the lane's output channels, loaded once (inside the epilogue it is a global load per work item whose latency every wave
  // waits out in front of its stores: conv_bf16x3_ws.h)
  float bvs[NBT];
#pragma unroll
  for (int j = 0; j < NBT; ++j) {
    const int co = n0 + j * 16 + lr;
    bvs[j] = (bias && !ksp && co < g.Cout) ? bias[co] : 0.f;
    ICL_PIN1(bvs[j]);
  }
  bool first_item = true;
  int item_no = -1;
  Origin nxt = {x, 0, 0, 0, 0, 0};
  Bf3RunStats<NBT> run;                  // InstanceNorm statistics of this workgroup's outputs (g.stats)
  run.reset();
  int my_sample = -1;
  while (tile < g.ntiles) {
    int ntile = tile, nchunk = chunk + 1;
    if (nchunk == c_hi) { nchunk = c_lo; ntile = next_tile(tile); }
    ++item_no;
    BF3_STAMP(0);
    __syncthreads();                       // everyone has finished reading the previous halo tile and weight plane
    BF3_STAMP(1);
#if !defined(BF3_DEBUG) || !(BF3_DEBUG & 2)
    store_x();
#endif
    if (WHOLE && (c_n > 1 || first_item)) store_w();
    first_item = false;
    BF3_STAMP(2);
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      if (!WHOLE) {
        if (dz > 0) __syncthreads();         // the previous plane's weights are no longer read
        store_w();
        __syncthreads();
        // global loads one phase ahead: the next weight plane first (older in the in-order vmcnt queue: waiting for it leaves the
        // halo tile of the next work item in flight), then — once per work item — that halo tile
        if (dz < 2) load_w(chunk, dz + 1);
        else if (ntile < g.ntiles) load_w(nchunk, 0);
      } else if (dz == 0) {
        __syncthreads();
        BF3_STAMP(3);
        if (c_n > 1 && ntile < g.ntiles) load_w(nchunk, 0);
      }
      // the halo loads of the next work item are not issued here as one burst (in-kernel stamps: 2.7-4.1k of a 20k-cycle item during
      // which both waves of a SIMD compute addresses and the matrix pipe idles) but one staging round per tap pair of the dz = 0
      // stage, inside that pair's scheduling region (buffer loads: no branches; past the last tile the current one is loaded again
      // and never stored)
      if (dz == 0) nxt = origin_of(ntile < g.ntiles ? ntile : tile, ntile < g.ntiles ? nchunk : chunk);
      if (dz == 0) BF3_STAMP(4);
      // stage dz holds the tap slots 10 dz .. 10 dz + 9 (27 taps + one zero slot = 14 pairs in stages of 5 / 5 / 4: a pair may
      // straddle two dz planes); lane group tp takes the first or the second tap of the pair: its halo offset is a select
      {
        // A pair's 24 NBT products in two halves of 12 NBT: X = the a1 terms (a1 b3, a1 b2, a1 b1: one operand plane of the four row
        // blocks, 4 reads), Y = (a3 b1, a2 b2, a2 b1: two planes, 8 reads).  The reads of a half are issued in front of the MFMAs of the
        // half before it — Y of this pair under X, X and the weight fragments of the NEXT pair under Y — so a wave waits for LDS
        // only at the start of a barrier-free run of pairs; until round 3 every pair was "15 reads, wait, 24 MFMAs" and the matrix
        // pipe idled whenever both waves of a SIMD were reading (multiply phase alone: 56 % busy, BF3_DEBUG = 6).
        const int np = dz < 2 ? 5 : 4;
        auto frag_ptr = [&](int sdz, int spair) {
          const int tA = 10 * sdz + 2 * spair, tB = tA + 1 < 27 ? tA + 1 : 26;
          const int offA = (tA / 9) * PY * PX + ((tA / 3) % 3) * PX + tA % 3, offB = (tB / 9) * PY * PX + ((tB / 3) % 3) * PX + tB % 3;
          return xa + (tp ? offB : offA);
        };
        auto load_b = [&](int buf, int sdz, int spair, int s0 = 0, int s1 = 3) {
#pragma unroll
          for (int s = s0; s < s1; ++s)
#pragma unroll
            for (int j = 0; j < NBT; ++j)
#if defined(BF3_DEBUG) && (BF3_DEBUG & 8)
              pb[buf][s][j] = make_uint4(sdz + s, spair, j, lane);
#else
              pb[buf][s][j] = wb[((WHOLE ? sdz * 6 : 0) * Bf3::SLOTS + s * 2 * Bf3::SLOTS + spair * 2) * NB + j * 16];
#endif
        };
        auto load_x1 = [&](int sdz, int spair) {
          const uint4* xp = frag_ptr(sdz, spair);
#pragma unroll
#if defined(BF3_DEBUG) && (BF3_DEBUG & 8)
          for (int m = 0; m < MB; ++m) pa1[m] = make_uint4(m, lane, (unsigned)(size_t)xp, 1u);
#else
          for (int m = 0; m < MB; ++m) pa1[m] = xp[moff[m]];
#endif
        };
        auto load_x23 = [&](int sdz, int spair) {
          const uint4* xp = frag_ptr(sdz, spair);
#pragma unroll
#if defined(BF3_DEBUG) && (BF3_DEBUG & 8)
          for (int m = 0; m < MB; ++m) { pa23[m][1] = make_uint4(m, lane, (unsigned)(size_t)xp, 2u); pa23[m][0] = make_uint4(m, lane, (unsigned)(size_t)xp, 3u); }
#else
          for (int m = 0; m < MB; ++m) pa23[m][1] = xp[4 * NPOSP + moff[m]];      // a3 first: its products lead the Y half
#pragma unroll
          for (int m = 0; m < MB; ++m) pa23[m][0] = xp[2 * NPOSP + moff[m]];
#endif
        };
        if (!WHOLE || dz == 0) {              // start of a barrier-free run: nothing was fetched ahead
          load_b(PIPE_B2 ? (5 * dz) & 1 : 0, dz, 0);
          load_x1(dz, 0);
        }
#pragma unroll
        for (int pair = 0; pair < np; ++pair) {
          const int cur = PIPE_B2 ? (5 * dz + pair) & 1 : 0;
          if (dz == 0 && pair < ROUNDS) load_x_at(nxt, pair, pair + 1);
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
            // in the order of first use by the next X half (b3, a1 of the four row blocks, b2, b1): woven between this half's MFMAs,
            // every read is then issued 11-12 MFMAs before the product that needs it
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
          BF3_STAMP(5 + 5 * dz + pair);
        }
      }
    }
    BF3_STAMP(20);
    if (chunk == c_hi - 1) {
      // ---- epilogue: lane holds x = 4 lq + r of row (wid, m), column co = n0 + 16 j + lr
      const int b = tile / tiles_per, bt = tile % tiles_per;
      const int x0 = (bt % g.ntx) * TC::TX, y0 = ((bt / g.ntx) % g.nty) * TC::TY, z0 = (bt / (g.ntx * g.nty)) * TC::TZ;
      float* yb = yout + (long)b * yout_bstride;
      const bool nt = (g.flags & 2) != 0;
#pragma unroll
      for (int j = 0; j < NBT; ++j) {
        const int co = n0 + j * 16 + lr;
        const float bv = bvs[j];
        float sv[16];
        bool sok[4] = {false, false, false, false};
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          int gz, gy, gx;
          if (FLAT) {      // the four positions 4 lq .. 4 lq + 3 of block 3 wid + m lie in one row (24 and 16 are multiples of 4)
            const int p = 16 * (MB * wid + m) + 4 * lq;
            gz = z0 + p / (TC::TY * TC::TX); gy = y0 + (p / TC::TX) % TC::TY; gx = x0 + p % TC::TX;
          } else {
            gz = z0 + wz; gy = y0 + wy + m; gx = x0 + 4 * lq;
          }
          const float4 v = make_float4(acc[m][j][0] + bv, acc[m][j][1] + bv, acc[m][j][2] + bv, acc[m][j][3] + bv);
          sv[4 * m] = v.x; sv[4 * m + 1] = v.y; sv[4 * m + 2] = v.z; sv[4 * m + 3] = v.w;
          sok[m] = co < g.Cout && gz < g.D && gy < g.H && gx < g.W;
#if defined(BF3_DEBUG) && (BF3_DEBUG & 4)
          if (acc[m][j][0] == 12345.678f)
#endif
          if (sok[m]) {
            float* dst = yb + (long)co * DHW + gz * HW + (long)gy * g.W + gx;
            if (nt) icl_nt_store4(dst, v); else *reinterpret_cast<float4*>(dst) = v;
          }
          acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (g.stats && !ksp) bf3_stats_add(run, j, sv, sok, 4 * MB);
      }
      my_sample = b;
    }
    BF3_STAMP(21);
    tile = ntile;
    chunk = nchunk;
  }
  if (g.stats && !ksp)
    bf3_stats_flush<NBT, TC::NW>(run, reinterpret_cast<float*>(Ws + WPL * TC::ws_u4(NB)), g.stats, my_sample, g.nbatch, g.Cout, n0, g.wgs,
                                 (int)blockIdx.x, wid, lane, tid);
}

// ------------------------------------------------------------------------------------------------ weight gradient, split products
// Geometry of the split-product weight gradient (conv_wgrad_tr.h: transposing LDS reads, two LDS buffers, eight waves).  The first form
// of round 2 (conv3d_bf16x3_wgrad_kernel: x-major LDS images, v_alignbit tap shifts, all 27 taps per wave in AGPRs; 1.1-1.4x the fp32
// kernels, 0.67x the round-3 form) lived here until round 5; its measurements are in DESIGN.md §4 and profiles/r2_pmc_conv.md.
struct Bf3WGeom {
  int Cin, Cout, CinP, CoutP, D, H, W;
  int ntz, nty, ntx, ntiles;      // tiles per sample; ntiles = batch * ntz * nty * ntx
  int tiles_per_wg;               // contiguous run of tiles per workgroup (blockIdx.x)
  long x_bstride, gy_bstride;
  int dbg;                        // timing ablations of conv_wgrad_tr.h (ICL_WGRAD_TR_DBG; results are wrong when set): 1 no global
                                  // loads, 2 no split / LDS stores, 4 no multiply
};

}  // namespace icl
