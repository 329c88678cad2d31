// device_env_hip.h — the gfx950 device environment the kernels in kernels/*.h are written against.
// (tests/hipemu/hipemu.h is a CPU-fiber implementation of the same names, used by tests only.)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x4_f32 — exact f32 (k-ordered fmaf chain), 32-cycle issue per SIMD.
// lane l: A[row l&15][k l>>4], B[k l>>4][col l&15]; D[row (l>>4)*4+r][col l&15] in reg r.
__device__ __forceinline__ f32x4 icl_mfma_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x2_f32 — lane l: A[row l&31][k l>>5], B[k l>>5][col l&31];
// D[row (r&3)+8*(r>>2)+4*(l>>5)][col l&31] in reg r.
__device__ __forceinline__ f32x16 icl_mfma_32x32x2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_16x16x32_bf16 — 16 cycles per SIMD.  Operands as 8 packed bf16 (uint4; low half of a dword = even element):
// lane l: A[row l&15][k 8*(l>>4) .. +7], B[k 8*(l>>4) .. +7][col l&15]; D as icl_mfma_16x16x4.  bf16 x bf16 products are exact in
// fp32; they are accumulated in fp32.
typedef __bf16 icl_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 icl_mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(icl_bf16x8, a), __builtin_bit_cast(icl_bf16x8, b), c, 0, 0, 0);
}

// v_cvt_pk_bf16_f32: two fp32 -> two bf16, round to nearest even (NaN stays NaN), packed {hi, lo} into one dword
typedef __bf16 icl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float icl_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned icl_pack_bf16_rn(float lo, float hi) {
  const icl_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, icl_bf16x2));
}

// ds_read_b64_tr_b16: transposing LDS read of 16-bit elements.  Within each group of 16 consecutive lanes, lane 4q + p supplies the
// address of row q, columns 4p .. 4p + 3 (8 bytes) of a 4-row x 16-column block; lane i receives column i of the four rows, row q in
// element q (cdna_hip_programming.md T10).  EXEC must be all ones.  Returns the four 16-bit elements as two dwords.
typedef short icl_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 icl_lds_read_tr16_b64(const void* lds_ptr) {
  typedef __attribute__((address_space(3))) icl_s16x4 lds_s16x4;
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lds_ptr));
}

// buffer_load_dword through a 128-bit resource descriptor (base pointer + byte extent held in scalar registers): 32-bit lane
// offset, and the hardware returns 0 for any offset outside the extent — a lane that must read nothing is handed an offset of
// 2^31 instead of a clamped address plus a select.  Build the descriptor from wave-uniform values only (kernel arguments, blockIdx).
typedef __amdgpu_buffer_rsrc_t icl_rsrc_t;
__device__ __forceinline__ icl_rsrc_t icl_make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float icl_buffer_load_f32(icl_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ uint4 icl_buffer_load_u32x4(icl_rsrc_t r, unsigned byte_off) {      // 16 bytes; every dword range-checked
  // (the whole vector is bit-cast at once: element-wise casts of the builtin's result make hipcc 7.2 narrow the load to ONE dword)
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  const u32x4v v = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 icl_buffer_load_u32x4(icl_rsrc_t r, unsigned byte_off, unsigned uniform_off) {
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  const u32x4v v = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, (int)uniform_off, 0));
  return make_uint4(v.x, v.y, v.z, v.w);
}
// ... with a wave-uniform byte offset on top (the instruction's scalar offset operand: no VALU add per load).  An out-of-range
// lane offset (>= the descriptor's extent) returns 0 whatever the scalar part is.
__device__ __forceinline__ float icl_buffer_load_f32(icl_rsrc_t r, unsigned byte_off, unsigned uniform_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)uniform_off, 0));
}

// buffer_load_dwordx4 ... offen lds (LDS-DMA): lane l's 16 bytes at descriptor offset byte_off (+ the wave-uniform uniform_off) are
// written to LDS at lds_wave_base + 16 l — no VGPR destination, no ds_write; the destination base must be wave-uniform (it travels in
// M0), only the SOURCE is per lane.  Counted by vmcnt like a load: the data are in LDS for other waves after the issuing wave's
// `s_waitcnt vmcnt` AND a barrier (ICL_WAIT_DMA + __syncthreads()).  Lanes outside the extent are handled by the caller (they are
// pointed at a zero block inside the extent): nothing here relies on what the hardware writes for a range-checked lane.
__device__ __forceinline__ void icl_buffer_load_lds_b128(icl_rsrc_t r, void* lds_wave_base, unsigned byte_off, unsigned uniform_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)byte_off, (int)uniform_off, 0, 0);
}
// every vector-memory operation of this wave (loads, stores, LDS-DMA) has completed
#define ICL_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// ... all but the n youngest (n a literal): loads, stores and LDS-DMA of a wave retire in issue order
#define ICL_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// Workgroup barrier that leaves vector-memory operations (LDS-DMA of a later stage, register loads) IN FLIGHT: __syncthreads() makes
// hipcc wait vmcnt(0) while an LDS-DMA is outstanding (cdna_hip_programming.md "Pipelining across barriers").  The wave's own LDS
// operations are complete before it arrives (lgkmcnt(0)); LDS-DMA data must have been waited for with ICL_WAIT_VMCNT by its issuer.
#define ICL_BARRIER_KEEP_VMEM()                          \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    asm volatile("" ::: "memory");                       \
  } while (0)

// v_alignbit_b32: bits [sh, sh + 32) of the 64-bit value {hi, lo}
__device__ __forceinline__ unsigned icl_alignbit(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// v_med3_f32: the median of three (one instruction).  med3(t, 0, +inf) = max(t, 0); med3(t, 0, 0) = 0: a clamp whose upper bound
// doubles as a per-lane "this position is padding" switch (kernels/conv_cl16.h)
__device__ __forceinline__ float icl_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// v_exp_f32 based exp (2 instructions); the CPU emulation maps it to expf
__device__ __forceinline__ float icl_fast_exp(float x) { return __expf(x); }

// hide an integer from constant folding: keeps LDS offsets small enough for ds_read2_b32 pairing (8-bit dword offsets)
#define ICL_OPAQUE_INT(x) asm volatile("" : "+v"(x))
// the four dwords of a uint4 are computed (and live in VGPRs) at this point of the program: stops the compiler from sinking a
// computation whose result is only stored much later
#define ICL_PIN4(u) asm volatile("" : "+v"((u).x), "+v"((u).y), "+v"((u).z), "+v"((u).w))
// ... one float: it is loaded and in its register here (the compiler's s_waitcnt for it sits at this point, not at a later use)
#define ICL_PIN1(f) asm volatile("" : "+v"(f))
// s_setprio: issue priority of this wave among the waves of its SIMD (0 lowest .. 3)
#define ICL_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
// nothing may be scheduled across this point (keeps software-prefetched LDS reads ahead of the MFMAs they overlap)
#define ICL_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// scheduling groups inside one region (LLVM AMDGPU masks: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x020 VMEM read, 0x002 VALU)
#define ICL_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// All lanes of a wave have executed what precedes this point before any lane continues.  On the GPU a wave runs in lockstep and
// its LDS operations execute in issue order, so this is only a compiler scheduling fence (no instruction); the CPU emulation,
// where lanes are independent fibers, makes it a rendezvous.  Used between the writes and reads of a wave-private LDS buffer.
#define ICL_WAVE_SYNC() __builtin_amdgcn_wave_barrier()

// tell the compiler a value is the same in every lane of the wave (lets it use SGPRs / scalar loads for what depends on it)
#define ICL_WAVE_UNIFORM(x) ((x) = __builtin_amdgcn_readfirstlane(x))

// non-temporal 16-byte accesses for streams that are touched once per launch (optimiser state): keep them out of L2 / MALL
__device__ __forceinline__ float4 icl_nt_load4(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void icl_nt_store4(float* p, float4 v) {
  const f32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p));
}

#define ICL_DYN_LDS(type, name)                                              \
  extern __shared__ __attribute__((aligned(16))) unsigned char icl_dyn_lds_raw[]; \
  type* name = reinterpret_cast<type*>(icl_dyn_lds_raw)

#define ICL_LAUNCH(kern, grid, block, lds, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), (lds), (stream), __VA_ARGS__)
#define ICL_MEMSET_ASYNC(ptr, val, bytes, stream) ((void)hipMemsetAsync((ptr), (val), (bytes), (stream)))
// address of a __device__ variable (0 on success)
#define ICL_SYMBOL_ADDRESS(pp, sym) ((int)hipGetSymbolAddress((pp), HIP_SYMBOL(sym)))
#define ICL_LAST_LAUNCH_ERROR() ((int)hipGetLastError())
#define ICL_ERROR_STRING(e) hipGetErrorString((hipError_t)(e))
// dynamic LDS above the 64 KiB default needs an explicit opt-in per kernel (160 KiB per CU on gfx950).  Done once per
// call site (= per template instantiation): it is not a stream operation and must stay out of hipGraph capture.
#define ICL_SET_MAX_DYN_LDS(kern, bytes)                                                                             \
  do {                                                                                                               \
    static bool icl_once_ = false;                                                                                   \
    if (!icl_once_) {                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
      icl_once_ = true;                                                                                              \
    }                                                                                                                \
  } while (0)
