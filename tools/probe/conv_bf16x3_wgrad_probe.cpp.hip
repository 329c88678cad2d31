// Standalone check + timing of the split-product weight-gradient kernel (csrc/kernels/conv_bf16x3.h):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icl_amd/csrc tools/probe/conv_bf16x3_wgrad_probe.cpp.hip -o tools/probe/bf3wprobe
//   tools/probe/bf3wprobe <cin> <cout> <side> [workgroups per block pair]
#include "device_env_hip.h"
#include "kernels/common.h"
#include "kernels/conv_bf16x3.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16, R = argc > 3 ? atoi(argv[3]) : 96, N = 2;
  const int D = R, H = R, W = R;
  const long S = (long)D * H * W;
  std::vector<float> hx((size_t)N * cin * S), hg((size_t)N * cout * S);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f + ((s >> 24) / 256.0f) * 1e-3f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hg) v = rnd() * 0.01f;
  icl::Bf3WGeom g{};
  g.Cin = cin; g.Cout = cout; g.CinP = (cin + 15) / 16 * 16; g.CoutP = (cout + 15) / 16 * 16; g.D = D; g.H = H; g.W = W;
  const int ncbk = argc > 5 ? atoi(argv[5]) : (g.CoutP % 32 == 0 ? 2 : 1), tz = ncbk >= 2 ? 2 : 4;
  g.ntz = (D + tz - 1) / tz; g.nty = (H + 7) / 8; g.ntx = (W + 15) / 16; g.ntiles = N * g.ntz * g.nty * g.ntx;
  const int pairs = (g.CinP / 16) * ((g.CoutP + 16 * ncbk - 1) / (16 * ncbk));
  int nsplit = argc > 4 && atoi(argv[4]) > 0 ? atoi(argv[4]) : (256 / pairs > 0 ? 256 / pairs : 1);   // at most one workgroup per CU
  if (nsplit > g.ntiles) nsplit = g.ntiles;
  g.tiles_per_wg = (g.ntiles + nsplit - 1) / nsplit;
  nsplit = (g.ntiles + g.tiles_per_wg - 1) / g.tiles_per_wg;
  g.x_bstride = cin * S; g.gy_bstride = cout * S;
  const long pe = 27L * g.CinP * g.CoutP;
  float *dx, *dg, *dslab;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dg, hg.size() * 4)); CK(hipMalloc(&dslab, (size_t)nsplit * pe * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dslab, 0xff, (size_t)nsplit * pe * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_wgrad_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)icl::Bf3WT<1, 4>::LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_wgrad_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)icl::Bf3WT<2, 2>::LDS_BYTES));
  const size_t lds1 = icl::Bf3WT<1, 4>::LDS_BYTES, lds2 = icl::Bf3WT<2, 2>::LDS_BYTES;
  auto go = [&]() {
    if (ncbk == 2) hipLaunchKernelGGL((icl::conv3d_bf16x3_wgrad_kernel<2, 2>), dim3(nsplit, pairs), dim3(256), lds2, 0, dx, dg, dslab, g);
    else hipLaunchKernelGGL((icl::conv3d_bf16x3_wgrad_kernel<1, 4>), dim3(nsplit, pairs), dim3(256), lds1, 0, dx, dg, dslab, g);
  };
  go();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  std::vector<float> hs((size_t)nsplit * pe);
  CK(hipMemcpy(hs.data(), dslab, hs.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  int bad = 0;
  for (int t = 0; t < 40; ++t) {
    s = s * 1664525u + 1013904223u;
    const int co = (s >> 8) % cout, ci = (s >> 16) % cin;
    int tap = t < 27 ? t : (s >> 24) % 27;
    const int dz = tap / 9, dy = (tap / 3) % 3, dxx = tap % 3;
    double ref = 0;
    for (int n = 0; n < N; ++n)
      for (int z = 0; z < D; ++z)
        for (int y = 0; y < H; ++y) {
          const int iz = z + dz - 1, iy = y + dy - 1;
          if (iz < 0 || iz >= D || iy < 0 || iy >= H) continue;
          const float* gp = &hg[((size_t)n * cout + co) * S + (size_t)z * H * W + (size_t)y * W];
          const float* xp = &hx[((size_t)n * cin + ci) * S + (size_t)iz * H * W + (size_t)iy * W];
          for (int xx = 0; xx < W; ++xx) {
            const int ix = xx + dxx - 1;
            if (ix < 0 || ix >= W) continue;
            ref += (double)gp[xx] * xp[ix];
          }
        }
    double got = 0;
    for (int k = 0; k < nsplit; ++k) got += hs[(size_t)k * pe + ((size_t)tap * g.CinP + ci) * g.CoutP + co];
    const double e = fabs(got - ref);
    if (!(e < 1e-3 * (fabs(ref) + 1.0))) { if (bad++ < 6) printf("  mismatch co %d ci %d tap %d: got %g ref %g\n", co, ci, tap, got, ref); }
    if (e > maxerr) maxerr = e;
    if (fabs(ref) > maxref) maxref = fabs(ref);
  }
  printf("wgrad %d->%d @%d^3 n=%d, %d slabs x %d block pairs: max abs err %.3e (max |ref| %.3f, rel %.2e), %d bad of 40\n", cin, cout, R, N, nsplit,
         pairs, maxerr, maxref, maxerr / maxref, bad);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) go();
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) go();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, fl = 2.0 * 27 * cin * cout * N * S;
  printf("  %.1f us per launch, %.1f fp32-equivalent TFLOP/s\n", us, fl / us * 1e-6);
  return bad ? 1 : 0;
}
