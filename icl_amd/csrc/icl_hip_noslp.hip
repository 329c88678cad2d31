// icl_hip_noslp.hip — second gfx950 translation unit of libicl_hip.so, compiled with -fno-slp-vectorize (icl_amd/build.py).
//
// Holds the kernels whose vector arithmetic runs BESIDE another wave's MFMA stream on the same SIMD.  hipcc's SLP vectoriser turns the
// residual subtractions of the three-way bf16 operand split (kernels/conv_bf16x3.h bf3_split2) into `v_pk_add_f32`; a packed f32
// instruction beside MFMAs costs its SIMD more than the two plain ones it replaces (MI355X_MICROARCH.md, cycle constants).  Measured on
// the loader-wave forward kernel (profiles/r6_pk_add_ab.txt, whole-library builds with and without the flag, interleaved, three
// repetitions): kernel cycles 7.00 -> 6.40 M on 16->16 @96^3 (matrix pipe busy 66 -> 72 %), 48->16 @96^3 378 -> 363 us (-4 %), 16->16
// best-of-rounds 133 -> 123 us; every other split-product kernel is 0-5 % SLOWER without the packing (its split runs while its own wave
// multiplies nothing, where the instruction count is what matters) and stays in the main unit.  Per-function control of the vectoriser
// does not exist in hipcc 7.2, and forming the residuals with opaque `v_sub_f32` asm in the main unit made the kernel spill (112 B at its
// 168-register budget; the flag build needs 165 and none): hence a unit of its own.  Same arithmetic, bit-identical results.
#define ICL_SECOND_UNIT 1
#include "device_env_hip.h"
#include "kernels/common.h"
#include "kernels/conv_bf16x3.h"
#include "kernels/conv_bf16x3_ws.h"

namespace icl {
template __global__ void conv3d_bf16x3_fwd_ws_kernel<1, false, false>(const float*, const uint4*, const float*, float*, Bf3Geom);
}
