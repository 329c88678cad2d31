// One-thread kernel that writes the GPU's constant-rate clock (100 MHz) into a slot: a mark that can be captured into a hipGraph on any
// stream, for tools/critical_path.py (which stream reaches which point of the ICL step when, under graph replay).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probe/stamp.hip -o tools/probe/libstamp.so
#include <hip/hip_runtime.h>
__global__ void stamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }
extern "C" int probe_stamp(unsigned long long* dst, void* stream) {
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
  return (int)hipGetLastError();
}
