// Does the age of the loader waves matter?  conv3d_bf16x3_fwd_ws_kernel<1> with its four loader waves as the workgroup's youngest waves
// (shipped) against the same kernel with the loaders as the OLDEST waves (template flag LF), optionally at another loader priority.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWS_LOADER_PRIO=n] -I icl_amd/csrc tools/probe/conv_ws_age_probe.hip -o /tmp/agep && /tmp/agep 16 16 96
// Round 5 (profiles/r5_cl16_stage_a.md): with no loads in flight at all a youngest-wave loader needs ~10 k cycles to issue its first 60 VALU
// instructions behind the consumers' MFMA streams.
#define WS_STAMPS 1
#include "device_env_hip.h"
#include "kernels/common.h"
#include "kernels/conv_bf16x3.h"
#include "kernels/conv_bf16x3_ws.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = 16, R = argc > 3 ? atoi(argv[3]) : 96, rounds = argc > 4 ? atoi(argv[4]) : 7;
  const int N = 2, D = R, H = R, W = R;
  const long S = (long)D * H * W;
  std::vector<float> hx((size_t)N * cin * S), hw((size_t)cout * cin * 27), hb(cout);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f + ((s >> 24) / 256.0f) * 1e-3f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  const int coutP = 16, cinP = cin;
  const long wpn = (long)27 * cinP * coutP;
  std::vector<float> hwp(wpn, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < 27; ++t) hwp[((size_t)t * cinP + ci) * coutP + co] = hw[((size_t)co * cin + ci) * 27 + t];
  float *dx, *db, *dy0, *dy1, *dwp, *dst;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dy0, (size_t)N * cout * S * 4)); CK(hipMalloc(&dy1, (size_t)N * cout * S * 4));
  CK(hipMalloc(&dwp, wpn * 4)); CK(hipMemcpy(dwp, hwp.data(), wpn * 4, hipMemcpyHostToDevice));
  uint4* dws; const long items = (long)(cin / 16) * 3 * 2 * icl::Bf3::SLOTS * coutP; CK(hipMalloc(&dws, items * 3 * 16));
  hipLaunchKernelGGL(icl::conv_bf16x3_split_weights_kernel, dim3(64), dim3(256), 0, 0, dwp, dws, cinP, coutP, cin / 16);
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  icl::Bf3Geom g{};
  g.Cin = cin; g.Cout = cout; g.CinP = cinP; g.CoutP = coutP; g.D = D; g.H = H; g.W = W;
  g.ntz = (D + 3) / 4; g.nty = (H + 7) / 8; g.ntx = (W + 15) / 16;
  g.ntiles = N * g.ntz * g.nty * g.ntx;
  g.nchunks = cin / 16; g.x_bstride = cin * S; g.y_bstride = cout * S;
  const int gx = g.ntiles < 256 ? (g.ntiles + 7) / 8 * 8 : 256;
  CK(hipMalloc(&dst, (size_t)N * cout * gx * 3 * 4));
  g.stats = dst; g.nbatch = N; g.wgs = gx;
  const size_t lds = icl::Bf3T<8>::lds_bytes(1, 3);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_ws_kernel<1, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_ws_kernel<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto go_old = [&]() { hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_ws_kernel<1, false, false>), dim3(gx, 1), dim3(768), lds, 0, dx, dws, db, dy0, g); };
  auto go_new = [&]() { hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_ws_kernel<1, false, true>), dim3(gx, 1), dim3(768), lds, 0, dx, dws, db, dy1, g); };
  CK(hipMemset(dy0, 0xff, (size_t)N * cout * S * 4)); CK(hipMemset(dy1, 0xee, (size_t)N * cout * S * 4));
  go_old(); go_new();
  CK(hipDeviceSynchronize());
  std::vector<float> h0((size_t)N * cout * S), h1(h0.size());
  CK(hipMemcpy(h0.data(), dy0, h0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), dy1, h1.size() * 4, hipMemcpyDeviceToHost));
  const bool same = !memcmp(h0.data(), h1.data(), h0.size() * 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](auto&& fn) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 100.0;
  };
  std::vector<double> a, b;
  for (int r = 0; r <= rounds; ++r) { const double x = timed(go_old), y = timed(go_new); if (r) { a.push_back(x); b.push_back(y); } }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const double fl = 2.0 * 27 * cin * cout * N * S;
#if !defined(WS_LOADER_PRIO)
#define WS_LOADER_PRIO 0
#endif
  printf("%d->16 @%d^3 n=2 loader priority %d: outputs %s | loaders youngest (shipped) %.1f us %.1f TF | loaders oldest %.1f us %.1f TF = %.3fx\n", cin, R,
         WS_LOADER_PRIO, same ? "bit-identical" : "DIFFER", med(a), fl / med(a) * 1e-6, med(b), fl / med(b) * 1e-6, med(a) / med(b));
  for (int which = 0; which < 2; ++which) {
    if (which) go_new(); else go_old();
    CK(hipDeviceSynchronize());
    long long st[144];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(icl::g_ws_stamps), sizeof(st)));
    const char* nm = which ? "oldest" : "youngest";
    for (int w = 0; w < 2; ++w) {
      const long long* q = st + (w * 3 + 1) * 16;
      printf("  [%s] consumer wave %d item 3: wait-A %lld | dz0 %lld dz1 %lld dz2 %lld | epilogue %lld | wait-B %lld | total %lld\n", nm, 4 * w,
             q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[6] - q[0]);
    }
    const long long* q = st + (2 * 3 + 1) * 16;
    printf("  [%s] loader wave 8 item 3: wait-A %lld | issue %lld split-a %lld split-b %lld | wait-B %lld | deposit %lld | total %lld\n", nm,
           q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[6] - q[0]);
  }
  return same ? 0 : 1;
}
