// winattn.h — Swin window attention (shifted-window multi-head self-attention inside 7x7x7 windows) on the fp32 MFMA.
//
// Reference: WindowAttention.forward, /root/reference/code/networks/swinunetr_icl.py:727-750 —
//     attn = softmax((q * scale) @ k^T + relative_position_bias + shift_mask);   x = attn @ v
// for every (window, head): n <= 343 tokens; head dim DH = 16 in every stage of SwinUNETR (C / heads = 48/3 = ... = 16) and
// DH = 32 in the 2-D Swin-UNet (7x7 windows, n = 49, networks/swinunet_icl.py:120-155) — a template parameter.
// Nothing of size n x n ever reaches HBM: one workgroup owns one (window, head), keeps K and V (n x 16 each) in LDS and
// each wave keeps the full score row block of its 16 queries in registers (22 key blocks x 4 VGPRs).
//
// MFMA mapping (v_mfma_f32_16x16x4_f32; lane l = (lr = l & 15, lg = l >> 4); A[row lr][k lg], B[k lg][col lr],
// D[row 4*lg + r][col lr] in register r).  The k index of an MFMA step is only a label, so operand rows are chosen such
// that NO register transposes are needed between the two GEMMs:
//   S^T block = K_blk (16 keys x 16 dims) * Q_blk^T : step t uses dims 4*lg + t on both operands (one ds_read_b128 of K and
//               one 16-byte load of Q per lane) -> lane holds S^T[key 4*lg + r][query lr];
//   O block   = P (16 queries x 16 keys) * V_blk    : step r uses key 4*lg + r on both operands: A is exactly the score
//               register r of the lane, B = V[key 4*lg + r][dim lr] (ds_read_b32, conflict-free with row stride 20).
// The softmax of a query runs over the 4 registers x key blocks of a lane and the 4 lanes sharing lr (2 shuffles).
//
// Shift mask: instead of the dense [nW, n, n] 0/-100 tensor of compute_mask (:979-1016; 161 MB for the 48^3 stage) the
// kernel reads the region id of every token ([nW, n] int32) and adds -100 where the ids of query and key differ.
// Bias: [heads, n, npad] with npad = n rounded up to 16 and the pad columns = -1e30 (masks the padded keys for free).
//
// Backward recomputes the scores from the saved log-sum-exp in two kernels of 58-60 KB LDS each (two workgroups per CU):
// (a) window_attn_bwd_kv — one workgroup per (window, head), scale*Q and dO of the window in LDS, waves own key blocks
//     (S layout) -> dK, dV, no atomics;
// (b) window_attn_bwd_q_bias — waves own a (head, query block) and walk over a slice of the windows (K, V staged per
//     window, S^T layout): dQ of every window is written directly, and d(bias)[h, i, j] — the sum of dS over ALL windows of
//     the batch — stays in a register-resident 16 x n slab that is added to HBM once at the end.
#pragma once

namespace icl {

constexpr float kWaMaskAdd = -100.0f;
constexpr int kWaThreads = 512;   // 8 waves per workgroup: two per SIMD hide the exp / LDS latency between MFMA chains
constexpr int kWaWaves = kWaThreads / 64;
constexpr float kWaPad = -1.0e30f; // bias value of the padded key columns

// LDS row pitch in floats (DH dims + 4 pad): ds_read_b128 rows and ds_read_b32 columns are conflict-free for DH = 16 and 32
template <int DH> struct WaCfg { static constexpr int LD = DH + 4, Q4 = DH / 4, DT = DH / 16; };

struct WinAttnGeom {
  int B_, n, npad, heads, nW;      // B_ = batch * nW windows; window id of row b_ is b_ % nW (window_partition order)
  float scale;
};

// exp via v_exp_f32 (2^x): relative error ~1e-6 for |x| < 20, far inside the 1e-3 parity budget
__device__ __forceinline__ float wa_exp(float x) { return icl_fast_exp(x); }

// rows [0, n) <- src[row * row_stride + 0..DH-1] (optionally scaled), rows [n, npad) <- 0
template <int DH>
__device__ __forceinline__ void wa_stage_rows(float* dst, const float* __restrict__ src, long row_stride, int n, int npad, float mul) {
  constexpr int LD = WaCfg<DH>::LD, Q4 = WaCfg<DH>::Q4;
  for (int it = threadIdx.x; it < npad * Q4; it += blockDim.x) {
    const int row = it / Q4, q = it % Q4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < n) {
      v = *reinterpret_cast<const float4*>(src + row * row_stride + q * 4);
      v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
    }
    *reinterpret_cast<float4*>(dst + row * LD + q * 4) = v;
  }
}

// Region ids of one window into LDS (0 where unmasked / padding).  `mixed` (one LDS word, cleared by the caller before the
// barrier that precedes this call) is raised when the window holds more than one region: most shifted windows are interior
// ones with a single region (216 of 343 at the 48^3 stage), and for those the per-score compare/select — a third of the
// softmax's VALU work — is skipped altogether.
__device__ __forceinline__ void wa_stage_regions(int* rid, int* mixed, const int* __restrict__ regions, long win_off, int n, int npad) {
  const int first = regions ? regions[win_off] : 0;
  for (int i = threadIdx.x; i < npad; i += blockDim.x) {
    const int v = (regions && i < n) ? regions[win_off + i] : 0;
    rid[i] = v;
    if (regions && i < n && v != first) *mixed = 1;
  }
}

// The DH values a lane contributes to a DH-deep contraction: dims 16*u + 4*lg + t (u < DH/16, t < 4) — MFMA step 4u + t.
template <int DH> struct WaFrag {
  float v[DH / 4];
  __device__ __forceinline__ void load(const float* row, int lg, float mul = 1.f) {
#pragma unroll
    for (int u = 0; u < DH / 16; ++u) {
      const float4 q = *reinterpret_cast<const float4*>(row + u * 16 + lg * 4);
      v[4 * u] = q.x * mul; v[4 * u + 1] = q.y * mul; v[4 * u + 2] = q.z * mul; v[4 * u + 3] = q.w * mul;
    }
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) v[i] = 0.f;
  }
};

// acc += A . B over the DH dims held in two fragments (A rows = the lanes' lr of fragment a, B columns = lr of fragment b)
template <int DH>
__device__ __forceinline__ f32x4 wa_dot(const WaFrag<DH>& a, const WaFrag<DH>& b, f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}) {
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) acc = icl_mfma_16x16x4(a.v[i], b.v[i], acc);
  return acc;
}

// S^T block for (key block kb, the wave's query block): lane -> scores of query lr against keys kb*16 + 4*lg + r.
template <int DH>
__device__ __forceinline__ f32x4 wa_scores_t(const float* Ks, int kb, int lr, int lg, const WaFrag<DH>& qf, const float* brow,
                                             const int* rid, int rq, bool masked) {
  WaFrag<DH> kf;
  kf.load(Ks + (kb * 16 + lr) * WaCfg<DH>::LD, lg);
  // the relative-position bias enters as the accumulator of the MFMA chain (no separate adds)
  const float4 b4 = *reinterpret_cast<const float4*>(brow + kb * 16 + lg * 4);
  f32x4 acc = wa_dot<DH>(kf, qf, f32x4{b4.x, b4.y, b4.z, b4.w});
  if (masked) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (rid[kb * 16 + lg * 4 + r] != rq) acc[r] += kWaMaskAdd;
  }
  return acc;
}

// ... with the bias quad of the key block already in registers (fetched one key block ahead by the caller)
template <int DH>
__device__ __forceinline__ f32x4 wa_scores_t_b(const float* Ks, int kb, int lr, int lg, const WaFrag<DH>& qf, float4 b4, const int* rid,
                                               int rq, bool masked) {
  WaFrag<DH> kf;
  kf.load(Ks + (kb * 16 + lr) * WaCfg<DH>::LD, lg);
  f32x4 acc = wa_dot<DH>(kf, qf, f32x4{b4.x, b4.y, b4.z, b4.w});
  if (masked) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (rid[kb * 16 + lg * 4 + r] != rq) acc[r] += kWaMaskAdd;
  }
  return acc;
}

// bias[h][i][j] = table[index[i * idx_stride + j]][h] for j < n, kWaPad for n <= j < npad
// (relative_position_bias_table[relative_position_index[:n, :n]], swinunetr_icl.py:733-737; idx_stride = 343).
__global__ __launch_bounds__(256) void relpos_bias_gather_kernel(const float* __restrict__ table, const long* __restrict__ index,
                                                                 float* __restrict__ bias, int n, int npad, int heads, int idx_stride) {
  const long total = (long)heads * n * npad;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int j = (int)(it % npad);
    const int i = (int)((it / npad) % n);
    const int h = (int)(it / ((long)npad * n));
    bias[it] = j < n ? table[index[(long)i * idx_stride + j] * heads + h] : kWaPad;
  }
}

// dtable[t][h] = sum over the (i, j) with index[i, j] == t and over the window slices of dbias[slice][h][i][j], in a FIXED order: one
// wave per (t, h); `inv` lists the padded positions i * npad + j sorted by t (stable), `offs[t] .. offs[t + 1]` its entries; lane l
// takes entries l, l + 64, .. and adds the slices of an entry in order, then the xor-shuffle tree.  No atomics: the bias-table
// gradient is bit-reproducible run to run.  grid (ceil(table_rows / 4), heads), block 256.
// dbias[0][i] += dbias[1][i] + .. + dbias[chunks - 1][i], slices added in order (coalesced 16-byte accesses): the gather below then
// reads ONE slab — its accesses are scattered (the positions of a table row lie all over the [n, npad] matrix), and scattered over
// 16 slabs they cost 112 us per layer at stage 1 instead of ~10.  slab % 4 == 0.
__global__ __launch_bounds__(256) void relpos_bias_slab_sum_kernel(float* __restrict__ dbias, long slab, int chunks) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < slab; i += (long)gridDim.x * blockDim.x * 4) {
    float4 a = *reinterpret_cast<const float4*>(dbias + i);
    for (int c = 1; c < chunks; ++c) {
      const float4 b = *reinterpret_cast<const float4*>(dbias + c * slab + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *reinterpret_cast<float4*>(dbias + i) = a;
  }
}

__global__ __launch_bounds__(256) void relpos_bias_gather_sum_kernel(const float* __restrict__ dbias, const int* __restrict__ inv,
                                                                     const int* __restrict__ offs, float* __restrict__ dtable, int table_rows,
                                                                     int n, int npad, int heads, int chunks) {
  const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y;
  if (t >= table_rows) return;
  const long slab = (long)heads * n * npad;
  const float* src = dbias + (long)h * n * npad;
  float acc = 0.f;
  for (int e = offs[t] + lane; e < offs[t + 1]; e += 64) {
    const int pos = inv[e];
    for (int c = 0; c < chunks; ++c) acc += src[c * slab + pos];
  }
  acc = wave_sum(acc);
  if (lane == 0) dtable[(long)t * heads + h] = acc;
}

// grid = B_ * heads workgroups of kWaThreads threads; LDS = (2 * npad * (DH + 4) + npad) * 4 bytes.
// EXACT: the window has exactly NKB key blocks (343 tokens -> 22, 216 -> 14): the per-block guards compile away and the whole
// query-block body becomes straight-line code, so the bias loads and K/V operand reads of later key blocks are issued under
// the MFMAs of earlier ones (with the guards every key block was its own basic block: load, wait, four MFMAs).
template <int NKB, int DH, bool EXACT>
__global__ __launch_bounds__(kWaThreads) void window_attn_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                              const int* __restrict__ regions, float* __restrict__ out,
                                                              float* __restrict__ lse, WinAttnGeom g) {
  constexpr int LD = WaCfg<DH>::LD, DT = WaCfg<DH>::DT;
  ICL_DYN_LDS(float, lds);
  float* Ks = lds;
  float* Vs = Ks + g.npad * LD;
  int* rid = reinterpret_cast<int*>(Vs + g.npad * LD);
  const int b_ = blockIdx.x / g.heads, h = blockIdx.x % g.heads;
  const int C = g.heads * DH, nkb = g.npad / 16;
  const long rs = 3L * C;
  const float* base = qkv + (long)b_ * g.n * rs + h * DH;
  wa_stage_rows<DH>(Ks, base + C, rs, g.n, g.npad, 1.f);
  wa_stage_rows<DH>(Vs, base + 2 * C, rs, g.n, g.npad, 1.f);
  __shared__ int mixed;
  if (threadIdx.x == 0) mixed = 0;
  __syncthreads();
  wa_stage_regions(rid, &mixed, regions, (long)(b_ % g.nW) * g.n, g.n, g.npad);
  __syncthreads();
  const bool masked = regions != nullptr && mixed != 0;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  for (int qb = wid; qb < nkb; qb += kWaWaves) {
    const int query = qb * 16 + lr, qc = query < g.n ? query : g.n - 1;
    WaFrag<DH> qf;
    qf.load(base + (long)qc * rs, lg, g.scale);
    const int rq = rid[qc];
    const float* brow = bias + ((long)h * g.n + qc) * g.npad;
    f32x4 s[NKB];
    float m = -3.0e38f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (EXACT || kb < nkb) {
        s[kb] = wa_scores_t<DH>(Ks, kb, lr, lg, qf, brow, rid, rq, masked);
        m = fmaxf(m, fmaxf(fmaxf(s[kb][0], s[kb][1]), fmaxf(s[kb][2], s[kb][3])));
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (EXACT || kb < nkb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = wa_exp(s[kb][r] - m);
          s[kb][r] = p;
          l += p;
        }
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (lg == 0 && query < g.n) lse[((long)b_ * g.heads + h) * g.n + query] = m + logf(l);
    const float inv = 1.0f / l;
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (EXACT || kb < nkb) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < DT; ++t) o[t] = icl_mfma_16x16x4(s[kb][r], Vs[(kb * 16 + lg * 4 + r) * LD + t * 16 + lr], o[t]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = qb * 16 + lg * 4 + r;
      const float iv = __shfl(inv, lg * 4 + r, 64);   // lane (lr = 4*lg + r, lg = 0) holds 1/l of that query
      if (q < g.n) {
#pragma unroll
        for (int t = 0; t < DT; ++t) out[((long)b_ * g.n + q) * C + h * DH + t * 16 + lr] = o[t][r] * iv;
      }
    }
  }
}

// Forward with an ONLINE softmax over chunks of four key blocks (round 3): the wave keeps 16 score registers
// instead of 4 * NKB (88 for 343 tokens), so the kernel needs ~100 registers instead of 218 and FOUR waves per SIMD (two workgroups per
// CU, which the 58 KB of LDS always allowed) hide the exp / LDS latencies that two could not.  Per chunk: scores (16 MFMAs), chunk row
// maximum (two lane-group shuffles), rescale of the running sum and of the O accumulators by exp(m_old - m_new) — O rows are queries
// 4 lg + r, the factor of query q lives in the lanes with lr = q: four shuffles —, P V (16 MFMAs).  Same outputs as
// window_attn_fwd_kernel (out, lse = m + log l).
template <int DH>
__global__ __launch_bounds__(kWaThreads, 2) void window_attn_fwd_online_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                                               const int* __restrict__ regions, float* __restrict__ out,
                                                                               float* __restrict__ lse, WinAttnGeom g) {
  constexpr int LD = WaCfg<DH>::LD, DT = WaCfg<DH>::DT, CH = 4;
  ICL_DYN_LDS(float, lds);
  float* Ks = lds;
  float* Vs = Ks + g.npad * LD;
  int* rid = reinterpret_cast<int*>(Vs + g.npad * LD);
  const int b_ = blockIdx.x / g.heads, h = blockIdx.x % g.heads;
  const int C = g.heads * DH, nkb = g.npad / 16;
  const long rs = 3L * C;
  const float* base = qkv + (long)b_ * g.n * rs + h * DH;
  wa_stage_rows<DH>(Ks, base + C, rs, g.n, g.npad, 1.f);
  wa_stage_rows<DH>(Vs, base + 2 * C, rs, g.n, g.npad, 1.f);
  __shared__ int mixed;
  if (threadIdx.x == 0) mixed = 0;
  __syncthreads();
  wa_stage_regions(rid, &mixed, regions, (long)(b_ % g.nW) * g.n, g.n, g.npad);
  __syncthreads();
  const bool masked = regions != nullptr && mixed != 0;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  for (int qb = wid; qb < nkb; qb += kWaWaves) {
    const int query = qb * 16 + lr, qc = query < g.n ? query : g.n - 1;
    WaFrag<DH> qf;
    qf.load(base + (long)qc * rs, lg, g.scale);
    const int rq = rid[qc];
    const float* brow = bias + ((long)h * g.n + qc) * g.npad;
    float m = -3.0e38f, l = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias quads of a chunk are fetched (from L2) one chunk ahead
    float4 bq[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) bq[k] = *reinterpret_cast<const float4*>(brow + (k < nkb ? k : 0) * 16 + lg * 4);
    for (int c0 = 0; c0 < nkb; c0 += CH) {
      f32x4 s[CH];
      float4 bc[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        bc[k] = bq[k];
        const int nk = c0 + CH + k;
        bq[k] = *reinterpret_cast<const float4*>(brow + (nk < nkb ? nk : 0) * 16 + lg * 4);
      }
      float mc = -3.0e38f;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        if (c0 + k < nkb) {
          s[k] = wa_scores_t_b<DH>(Ks, c0 + k, lr, lg, qf, bc[k], rid, rq, masked);
          mc = fmaxf(mc, fmaxf(fmaxf(s[k][0], s[k][1]), fmaxf(s[k][2], s[k][3])));
        }
      }
      mc = fmaxf(mc, __shfl_xor(mc, 16, 64));
      mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
      const float mn = fmaxf(m, mc);
      const float alpha = wa_exp(m - mn);                 // 0 on the first chunk (m = -3e38), 1 when the maximum did not move
      l *= alpha;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        if (c0 + k < nkb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = wa_exp(s[k][r] - mn);
            s[k][r] = pv;
            l += pv;
          }
        }
      }
      // O row 4 lg + r belongs to query 4 lg + r: its factor sits in the lanes with lr = 4 lg + r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ar = __shfl(alpha, lg * 4 + r, 64);
#pragma unroll
        for (int t = 0; t < DT; ++t) o[t][r] *= ar;
      }
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        if (c0 + k < nkb) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < DT; ++t)
              o[t] = icl_mfma_16x16x4(s[k][r], Vs[((c0 + k) * 16 + lg * 4 + r) * LD + t * 16 + lr], o[t]);
        }
      }
      m = mn;
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (lg == 0 && query < g.n) lse[((long)b_ * g.heads + h) * g.n + query] = m + logf(l);
    const float inv = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = qb * 16 + lg * 4 + r;
      const float iv = __shfl(inv, lg * 4 + r, 64);
      if (q < g.n) {
#pragma unroll
        for (int t = 0; t < DT; ++t) out[((long)b_ * g.n + q) * C + h * DH + t * 16 + lr] = o[t][r] * iv;
      }
    }
  }
}

// dK and dV.  grid = B_ * heads; LDS = (2 * npad * (DH + 4) + 3 * npad) * 4 bytes (scale*Q, dO, log-sum-exp, delta, region ids of
// every query of the window); waves own key blocks, their K / V rows come straight from HBM as MFMA operands.
// S layout: lane -> query 4*lg + r of the block, key lr.  dqkv has the layout of qkv; only the k and v thirds are written here.
template <int DH>
__global__ __launch_bounds__(kWaThreads) void window_attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                                 const int* __restrict__ regions, const float* __restrict__ out,
                                                                 const float* __restrict__ lse, const float* __restrict__ dout,
                                                                 float* __restrict__ dqkv, WinAttnGeom g) {
  constexpr int LD = WaCfg<DH>::LD, DT = WaCfg<DH>::DT, Q4 = WaCfg<DH>::Q4;
  ICL_DYN_LDS(float, lds);
  const int np = g.npad;
  float* Qs = lds;                 // scale * Q
  float* Gs = Qs + np * LD;        // dO
  float* Ls = Gs + np * LD;        // log-sum-exp per query (+1e30 on pad rows -> p = 0)
  float* Ds = Ls + np;             // delta[q] = sum_dim dO * O
  int* rid = reinterpret_cast<int*>(Ds + np);
  const int b_ = blockIdx.x / g.heads, h = blockIdx.x % g.heads;
  const int C = g.heads * DH, nkb = np / 16;
  const long rs = 3L * C;
  const float* base = qkv + (long)b_ * g.n * rs + h * DH;
  const float* dob = dout + (long)b_ * g.n * C + h * DH;
  const float* ob = out + (long)b_ * g.n * C + h * DH;
  wa_stage_rows<DH>(Qs, base, rs, g.n, np, g.scale);
  wa_stage_rows<DH>(Gs, dob, C, g.n, np, 1.f);
  __shared__ int mixed;
  if (threadIdx.x == 0) mixed = 0;
  __syncthreads();
  wa_stage_regions(rid, &mixed, regions, (long)(b_ % g.nW) * g.n, g.n, np);
  for (int i = threadIdx.x; i < np; i += blockDim.x) {
    float dl = 0.f, ls = 1.0e30f;
    if (i < g.n) {
      ls = lse[((long)b_ * g.heads + h) * g.n + i];
#pragma unroll
      for (int q = 0; q < Q4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(dob + (long)i * C + q * 4);
        const float4 b = *reinterpret_cast<const float4*>(ob + (long)i * C + q * 4);
        dl += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      }
    }
    Ls[i] = ls;
    Ds[i] = dl;
  }
  __syncthreads();
  const bool masked = regions != nullptr && mixed != 0;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  for (int kb = wid; kb < nkb; kb += kWaWaves) {
    const int key = kb * 16 + lr, kc = key < g.n ? key : g.n - 1;
    WaFrag<DH> kf, vf;
    kf.load(base + C + (long)kc * rs, lg);
    vf.load(base + 2 * C + (long)kc * rs, lg);
    if (key >= g.n) { kf.zero(); vf.zero(); }
    const int rk = rid[kc];
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) { dk[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // the four bias values of a query block (rows 4 lg + r, column `key`) come from L2: fetched one query block ahead, and they
    // enter as the accumulator of the score MFMAs (round 3: they were loaded inside the dependent chain load -> add -> exp)
    const float* bcol = bias + (long)h * g.n * np + key;
    auto load_bias = [&](int qb) {
      f32x4 b;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = qb * 16 + lg * 4 + r, qc = q < g.n ? q : g.n - 1;
        b[r] = bcol[(long)qc * np];
      }
      return b;
    };
    f32x4 bnext = load_bias(0);
#pragma unroll 2
    for (int qb = 0; qb < nkb; ++qb) {
      const f32x4 bcur = bnext;
      if (qb + 1 < nkb) bnext = load_bias(qb + 1);
      WaFrag<DH> qf, gf;
      qf.load(Qs + (qb * 16 + lr) * LD, lg);
      gf.load(Gs + (qb * 16 + lr) * LD, lg);
      const f32x4 s = wa_dot<DH>(qf, kf, bcur);
      const f32x4 dp = wa_dot<DH>(gf, vf);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = qb * 16 + lg * 4 + r, qc = q < g.n ? q : g.n - 1;
        float sv = s[r];
        if (masked && rid[qc] != rk) sv += kWaMaskAdd;
        const float p = wa_exp(sv - Ls[q]);
        const float ds = p * (dp[r] - Ds[q]);
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          dv[t] = icl_mfma_16x16x4(p, Gs[q * LD + t * 16 + lr], dv[t]);
          dk[t] = icl_mfma_16x16x4(ds, Qs[q * LD + t * 16 + lr], dk[t]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kb * 16 + lg * 4 + r;
      if (k < g.n) {
        float* row = dqkv + ((long)b_ * g.n + k) * rs + h * DH + lr;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          row[C + t * 16] = dk[t][r];
          row[2 * C + t * 16] = dv[t][r];
        }
      }
    }
  }
}

// dQ and d(bias).  Waves own a (head, query block) and walk over a slice of the windows: for every window K and V are staged
// in LDS, dS is recomputed (S^T layout), dQ of the 16 queries is written, and dS is added to a register-resident 16 x n slab
// of d(bias), which is summed over all windows of the batch and added to HBM once at the end.
// grid (ceil(nkb/kWaWaves), heads, chunks); LDS = (2 * npad * (DH + 4) + npad) * 4 bytes.  dbias [chunks, heads, n, npad] (may be NULL).
template <int NKB, int DH, bool EXACT>
__global__ __launch_bounds__(kWaThreads) void window_attn_bwd_q_bias_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                                     const int* __restrict__ regions, const float* __restrict__ out,
                                                                     const float* __restrict__ lse, const float* __restrict__ dout,
                                                                     float* __restrict__ dqkv, float* __restrict__ dbias, WinAttnGeom g) {
  constexpr int LD = WaCfg<DH>::LD, DT = WaCfg<DH>::DT;
  ICL_DYN_LDS(float, lds);
  float* Ks = lds;
  float* Vs = Ks + g.npad * LD;
  int* rid = reinterpret_cast<int*>(Vs + g.npad * LD);
  const int h = blockIdx.y, C = g.heads * DH, nkb = g.npad / 16;
  const long rs = 3L * C;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int qb = blockIdx.x * kWaWaves + wid;
  const bool active = qb < nkb;
  const int query = qb * 16 + lr, qc = (active && query < g.n) ? query : g.n - 1;
  const bool qvalid = active && query < g.n;
  __shared__ int mixed;
  const float* brow = bias + ((long)h * g.n + qc) * g.npad;
  const int per = (g.B_ + gridDim.z - 1) / gridDim.z;
  const int b0 = blockIdx.z * per, b1 = (b0 + per < g.B_) ? b0 + per : g.B_;
  f32x4 acc[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) acc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int b_ = b0; b_ < b1; ++b_) {
    const float* base = qkv + (long)b_ * g.n * rs + h * DH;
    __syncthreads();
    if (threadIdx.x == 0) mixed = 0;
    wa_stage_rows<DH>(Ks, base + C, rs, g.n, g.npad, 1.f);
    wa_stage_rows<DH>(Vs, base + 2 * C, rs, g.n, g.npad, 1.f);
    __syncthreads();
    wa_stage_regions(rid, &mixed, regions, (long)(b_ % g.nW) * g.n, g.n, g.npad);
    __syncthreads();
    const bool masked = regions != nullptr && mixed != 0;
    if (!active) continue;
    WaFrag<DH> qf, gf, of;
    qf.load(base + (long)qc * rs, lg, g.scale);
    gf.load(dout + ((long)b_ * g.n + qc) * C + h * DH, lg);
    of.load(out + ((long)b_ * g.n + qc) * C + h * DH, lg);
    float dl = 0.f;
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) dl += gf.v[i] * of.v[i];
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float lq = qvalid ? lse[((long)b_ * g.heads + h) * g.n + qc] : 1.0e30f;
    const int rq = rid[qc];
    f32x4 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bq = *reinterpret_cast<const float4*>(brow + lg * 4);      // bias quad of key block 0; the next one is fetched under this one's work
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (EXACT || kb < nkb) {
        const float4 bcur = bq;
        if (kb + 1 < NKB && (EXACT || kb + 1 < nkb)) bq = *reinterpret_cast<const float4*>(brow + (kb + 1) * 16 + lg * 4);
        const f32x4 s = wa_scores_t_b<DH>(Ks, kb, lr, lg, qf, bcur, rid, rq, masked);
        WaFrag<DH> vf;
        vf.load(Vs + (kb * 16 + lr) * LD, lg);
        const f32x4 dp = wa_dot<DH>(vf, gf);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = wa_exp(s[r] - lq) * (dp[r] - dl);
          acc[kb][r] += ds;
#pragma unroll
          for (int t = 0; t < DT; ++t) dq[t] = icl_mfma_16x16x4(ds, Ks[(kb * 16 + lg * 4 + r) * LD + t * 16 + lr], dq[t]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = qb * 16 + lg * 4 + r;
      if (q < g.n) {
#pragma unroll
        for (int t = 0; t < DT; ++t) dqkv[((long)b_ * g.n + q) * rs + h * DH + t * 16 + lr] = dq[t][r] * g.scale;
      }
    }
  }
  if (!qvalid || !dbias) return;
  // this window slice's OWN slab dbias[blockIdx.z][h][query][.]: plain stores, every (query, key) written exactly once; the slabs are
  // summed in a fixed order by relpos_bias_gather_sum_kernel (round 4: the float atomics here were the last run-to-run non-determinism
  // of the SwinUNETR step)
  float* drow = dbias + (((long)blockIdx.z * g.heads + h) * g.n + query) * g.npad;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    if (EXACT || kb < nkb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = kb * 16 + lg * 4 + r;
        if (k < g.n) drow[k] = acc[kb][r];
      }
    }
  }
}

}  // namespace icl
