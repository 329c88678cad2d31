// common.h — small device helpers shared by all kernels (wave = 64 lanes on gfx950).
// Written against the device environment (device_env_hip.h); includes nothing itself.
#pragma once

namespace icl {

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// Sum over a whole block of NT threads (NT multiple of 64, <= 1024); result valid in every thread.
// `red` is an LDS scratch array of at least NT/64 floats owned by the caller.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous use
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

// Tickets of a split reduction (OPT-IN, ICL_TICKETS=1; measured slower on MI355X, see below).  Every workgroup that contributes a partial
// result to an output region stores it, then draws a ticket of the region's counter; the one that draws the LAST ticket (all partials are
// in memory by then) sums them — always in slab order, so the result does not depend on which workgroup that was — and zeroes the counter
// for the next launch.  The counter is the only atomic: an integer, it decides who adds, never what is added.  `flag` is one word of LDS
// that no wave uses any more once it gets here (the kernels pass the start of their dynamic LDS: a static __shared__ word would take a
// kernel over the 160 KiB of dynamic LDS it may ask for).
// Why it is off: the partials cross XCDs, so the hand-over needs device-scope release / acquire, and on gfx950 those are whole-cache
// operations — `buffer_wbl2 sc1` writes back every dirty line of the XCD's L2, `buffer_inv sc1` invalidates it, for every kernel running
// there.  U-Net ICL step 13.85 ms with __threadfence() in every thread against 12.08 with gemm_reduce_slabs_kernel as a second launch
// (SwinUNETR 54.1 against 46.5); with one release per workgroup and one acquire in the summing workgroup 12.32 against 11.96 (50.4 / 45.3)
// — and that form then failed the three-step golden test.  profiles/r4_tickets_ab.txt.
constexpr unsigned kTicketRing = 1u << 16;
__device__ unsigned g_ticket_ring[kTicketRing];      // zero at load; handed out in ranges by the launchers (icl_abi.inc tickets_take)

__device__ __forceinline__ bool ticket_is_last(unsigned* ticket, unsigned contributors, unsigned* flag) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket, 1u);
    *flag = t + 1u == contributors ? 1u : 0u;
    if (t + 1u == contributors) atomicExch(ticket, 0u);
  }
  __syncthreads();
  const bool last = *flag != 0u;
  if (last) __threadfence();
  return last;
}

// component j of a float4 (j is a compile-time constant after unrolling)
__device__ __forceinline__ float f4c(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

__device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Chan et al. merge of two (count, mean, M2) summaries — numerically stable variance.
__device__ __forceinline__ void welford_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb;
  const float delta = meanb - mean;
  const float f = nb / nt;
  mean = mean + delta * f;
  m2 = m2 + m2b + delta * delta * n * f;
  n = nt;
}

}  // namespace icl
