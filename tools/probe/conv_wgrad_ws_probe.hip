// Standalone A/B of tools/probe/probe_kernels/conv_wgrad_tr_ws.h (weight gradient with producer waves) against the shipped
// conv3d_wgrad_tr_kernel<NCB>: bitwise comparison of the partial-dW slabs, race screen, interleaved timing, in-kernel stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icl_amd/csrc -I tools/probe tools/probe/conv_wgrad_ws_probe.hip -o tools/probe/wgradwsprobe
//   tools/probe/wgradwsprobe 16 16 96 [rounds]
#define WGTR_WS_STAMPS 1
#include "device_env_hip.h"
#include "kernels/common.h"
#include "kernels/conv_bf16x3.h"
#include "kernels/conv_wgrad_tr.h"
#include "probe_kernels/conv_wgrad_tr_ws.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NCB>
int run(int cin, int cout, int R, int rounds) {
  typedef icl::WgTrT<NCB> C;
  const int N = 2, D = R, H = R, W = R;
  const long S = (long)D * H * W;
  std::vector<float> hx((size_t)N * cin * S), hg((size_t)N * cout * S);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f + ((s >> 24) / 256.0f) * 1e-3f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hg) v = rnd() * 0.01f;
  icl::Bf3WGeom g{};
  g.Cin = cin; g.Cout = cout; g.CinP = (cin + 15) / 16 * 16; g.CoutP = (cout + 15) / 16 * 16; g.D = D; g.H = H; g.W = W;
  g.ntz = (D + C::TZ - 1) / C::TZ; g.nty = (H + C::TY - 1) / C::TY; g.ntx = (W + C::TX - 1) / C::TX; g.ntiles = N * g.ntz * g.nty * g.ntx;
  const int pairs = ((g.CoutP + 16 * NCB - 1) / (16 * NCB)) * (g.CinP / 16);
  int nsplit = 256 / pairs > 0 ? 256 / pairs : 1;
  if (nsplit > g.ntiles) nsplit = g.ntiles;
  g.tiles_per_wg = (g.ntiles + nsplit - 1) / nsplit;
  nsplit = (g.ntiles + g.tiles_per_wg - 1) / g.tiles_per_wg;
  g.x_bstride = cin * S; g.gy_bstride = cout * S; g.dbg = 0;
  const long pe = 27L * g.CinP * g.CoutP;
  float *dx, *dg, *s0, *s1;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dg, hg.size() * 4));
  CK(hipMalloc(&s0, (size_t)nsplit * pe * 4)); CK(hipMalloc(&s1, (size_t)nsplit * pe * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_wgrad_tr_kernel<NCB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_wgrad_tr_ws_kernel<NCB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto go_old = [&]() { hipLaunchKernelGGL((icl::conv3d_wgrad_tr_kernel<NCB>), dim3(nsplit, pairs), dim3(512), C::LDS_BYTES, 0, dx, dg, s0, g); };
  auto go_new = [&]() { hipLaunchKernelGGL((icl::conv3d_wgrad_tr_ws_kernel<NCB>), dim3(nsplit, pairs), dim3(512 + 64 * WSW_PRODUCERS), C::LDS_BYTES, 0, dx, dg, s1, g); };
  CK(hipMemset(s0, 0xff, (size_t)nsplit * pe * 4)); CK(hipMemset(s1, 0xee, (size_t)nsplit * pe * 4));
  go_old(); go_new();
  CK(hipDeviceSynchronize()); CK(hipGetLastError());
  std::vector<float> h0((size_t)nsplit * pe), h1(h0.size()), h2(h0.size());
  CK(hipMemcpy(h0.data(), s0, h0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), s1, h1.size() * 4, hipMemcpyDeviceToHost));
  size_t ndiff = 0;
  for (size_t i = 0; i < h0.size(); ++i) if (memcmp(&h0[i], &h1[i], 4)) { if (ndiff++ < 5) printf("  diff at %zu: shipped %g ws %g\n", i, h0[i], h1[i]); }
  printf("[%d producer waves, priority %d] %d->%d @%d^3 n=%d NCB=%d, %d workgroups x %d tiles: %zu of %zu slab values differ bitwise\n", WSW_PRODUCERS, WSW_PRIO, cin, cout, R, N, NCB, nsplit * pairs, g.tiles_per_wg, ndiff, h0.size());
  for (int rep = 0; rep < 8; ++rep) {
    CK(hipMemset(s1, 0xee, (size_t)nsplit * pe * 4));
    go_new();
    CK(hipMemcpy(h2.data(), s1, h2.size() * 4, hipMemcpyDeviceToHost));
    if (memcmp(h2.data(), h1.data(), h1.size() * 4)) { printf("  RACE: repeat %d differs from the first run\n", rep); ++ndiff; }
  }
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
  for (int r = 0; r <= rounds; ++r) { const double x0 = timed(go_old), x1 = timed(go_new); if (r) { a.push_back(x0); b.push_back(x1); } }
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  const double fl = 2.0 * 27 * cin * cout * N * S;
  printf("  shipped <%d>: median %.1f us (min %.1f) %.1f TF | producer waves: median %.1f us (min %.1f) %.1f TF = %.3fx\n", NCB, a[a.size() / 2], a[0],
         fl / a[a.size() / 2] * 1e-6, b[b.size() / 2], b[0], fl / b[b.size() / 2] * 1e-6, a[a.size() / 2] / b[b.size() / 2]);
  long long st[96];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(icl::g_wgtr_ws_stamps), sizeof(st)));
  for (int w = 0; w < 3; ++w) {
    const long long* q = st + (w * 4 + 1) * 8;      // phase 5
    if (w < 2) printf("  stamps consumer wave %d phase 5: multiply %lld | wait at the barrier %lld | phase %lld\n", 4 * w, q[1] - q[0], q[4] - q[3], q[4] - q[0]);
    else printf("  stamps producer wave 8 phase 5: split + store %lld | issue loads %lld | wait at the barrier %lld | phase %lld\n", q[1] - q[0], q[2] - q[1], q[4] - q[3], q[4] - q[0]);
  }
  return ndiff ? 1 : 0;
}

int main(int argc, char** argv) {
  const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16, R = argc > 3 ? atoi(argv[3]) : 96;
  const int rounds = argc > 4 ? atoi(argv[4]) : 5;
  const int coutP = (cout + 15) / 16 * 16;
  const int force1 = argc > 5 ? atoi(argv[5]) : 0;      // 1: one cout block per workgroup column also for 32-multiples
  if (coutP % 32 == 0 && !force1) return run<2>(cin, cout, R, rounds);
  return run<1>(cin, cout, R, rounds);
}
