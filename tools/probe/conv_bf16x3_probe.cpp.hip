// Standalone check + timing of csrc/kernels/conv_bf16x3.h (3x3x3 convolution on bf16 x 3 split operands):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icl_amd/csrc tools/probe/conv_bf16x3_probe.cpp.hip -o /tmp/bf3probe && /tmp/bf3probe 16 16 96
// Prints the error against an fp64 CPU convolution on sampled outputs and the launch time / fp32-equivalent TFLOP/s.
#include "device_env_hip.h"
#include "kernels/common.h"
#include "kernels/conv_bf16x3.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static int g_wgs = 256;
template <int NBT, int TY>
void launch(const float* x, const uint4* wp, const float* bias, float* y, icl::Bf3Geom g, hipStream_t st) {
  const size_t lds = icl::Bf3T<TY>::lds_bytes(NBT);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_kernel<NBT, TY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int gy = (g.CoutP + 16 * NBT - 1) / (16 * NBT);
  g.nty = (g.H + TY - 1) / TY;
  g.ntiles = 2 * g.ntz * g.nty * g.ntx;
  int gx = g.ntiles < g_wgs ? (g.ntiles + 7) / 8 * 8 : g_wgs;
  hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_kernel<NBT, TY>), dim3(gx, gy), dim3(64 * TY), lds, st, x, wp, bias, y, g);
}

int main(int argc, char** argv) {
  const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16, R = argc > 3 ? atoi(argv[3]) : 96;
  const int nbt = argc > 4 ? atoi(argv[4]) : (cout % 48 == 0 ? 3 : cout % 32 == 0 ? 2 : 1), N = 2;
  const int ty = argc > 5 ? atoi(argv[5]) : 8;
  g_wgs = argc > 6 ? atoi(argv[6]) : (ty == 4 ? 512 : 256);
  const int D = R, H = R, W = R;
  const long S = (long)D * H * W;
  std::vector<float> hx((size_t)N * cin * S), hw((size_t)cout * cin * 27), hb(cout);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f + ((s >> 24) / 256.0f) * 1e-3f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  float *dx, *dw, *db, *dy;
  float* dwp;
  const int coutP = (cout + 15) / 16 * 16, cinP = cin;
  const long wpn = (long)27 * cinP * coutP;
  std::vector<float> hwp(wpn, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < 27; ++t) hwp[((size_t)t * cinP + ci) * coutP + co] = hw[((size_t)co * cin + ci) * 27 + t];
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dy, (size_t)N * cout * S * 4)); CK(hipMalloc(&dwp, wpn * 4)); CK(hipMemcpy(dwp, hwp.data(), wpn * 4, hipMemcpyHostToDevice));
  uint4* dws; const long items = (long)(cin / 16) * 3 * 2 * icl::Bf3::SLOTS * coutP; CK(hipMalloc(&dws, items * 3 * 16));
  hipLaunchKernelGGL(icl::conv_bf16x3_split_weights_kernel, dim3(64), dim3(256), 0, 0, dwp, dws, cinP, coutP, cin / 16);
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0xff, (size_t)N * cout * S * 4));
  icl::Bf3Geom g{};
  g.Cin = cin; g.Cout = cout; g.CinP = cinP; g.CoutP = coutP; g.D = D; g.H = H; g.W = W;
  g.ntz = (D + 3) / 4; g.nty = (H + 7) / 8; g.ntx = (W + 15) / 16; g.ntiles = N * g.ntz * g.nty * g.ntx;
  g.nchunks = cin / 16; g.x_bstride = cin * S; g.y_bstride = cout * S;
  auto go = [&]() {
    if (ty == 4) { if (nbt == 1) launch<1, 4>(dx, dws, db, dy, g, 0); else if (nbt == 2) launch<2, 4>(dx, dws, db, dy, g, 0); else launch<3, 4>(dx, dws, db, dy, g, 0); }
    else { if (nbt == 1) launch<1, 8>(dx, dws, db, dy, g, 0); else if (nbt == 2) launch<2, 8>(dx, dws, db, dy, g, 0); else launch<3, 8>(dx, dws, db, dy, g, 0); }
  };
  go();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  std::vector<float> hy((size_t)N * cout * S);
  CK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  int bad = 0;
  for (int t = 0; t < 4000; ++t) {
    s = s * 1664525u + 1013904223u;
    const int n = (s >> 4) % N, co = (s >> 8) % cout;
    s = s * 1664525u + 1013904223u;
    int z = (s >> 3) % D, yy = (s >> 11) % H, xx = (s >> 19) % W;
    if (t < 64) { z = (t & 1) ? D - 1 : 0; yy = (t & 2) ? H - 1 : 0; xx = (t & 4) ? W - 1 : (t & 8) ? 15 : 0; }
    double ref = hb[co];
    for (int ci = 0; ci < cin; ++ci)
      for (int dz = 0; dz < 3; ++dz)
        for (int dy_ = 0; dy_ < 3; ++dy_)
          for (int dx_ = 0; dx_ < 3; ++dx_) {
            const int iz = z + dz - 1, iy = yy + dy_ - 1, ix = xx + dx_ - 1;
            if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            ref += (double)hw[((size_t)co * cin + ci) * 27 + dz * 9 + dy_ * 3 + dx_] * hx[((size_t)n * cin + ci) * S + (size_t)iz * H * W + iy * W + ix];
          }
    const double got = hy[((size_t)n * cout + co) * S + (size_t)z * H * W + yy * W + xx];
    const double e = fabs(got - ref);
    if (!(e < 1e-3)) { if (bad++ < 5) printf("  mismatch n %d co %d z %d y %d x %d: got %g ref %g\n", n, co, z, yy, xx, got, ref); }
    if (e > maxerr) maxerr = e;
    if (fabs(ref) > maxref) maxref = fabs(ref);
  }
  printf("%d->%d @%d^3 n=%d nbt=%d ty=%d: max abs err %.3e (max |ref| %.3f, rel %.2e), %d bad of 4000\n", cin, cout, R, N, nbt, ty, maxerr, maxref, maxerr / maxref, bad);
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
  printf("  %.1f us per launch, %.1f fp32-equivalent TFLOP/s (%.1f TFLOP/s of bf16 MFMA work)\n", us, fl / us * 1e-6, 6 * fl / us * 1e-6 * 30 / 27);
  return bad ? 1 : 0;
}
