// Standalone A/B of kernels/conv_wgrad_zs.h (weight gradient walking z-columns of 1 x 8 x 16 tiles with a ring of halo planes) against
// conv3d_wgrad_tr_kernel<NCB>: comparison of the summed dW (the tiles differ, so the slabs are not bitwise comparable), race screen,
// interleaved timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icl_amd/csrc -I tools/probe tools/probe/conv_wgrad_zs_probe.hip -o tools/probe/wgradzsprobe
//   tools/probe/wgradzsprobe 16 16 96 [rounds] [force one cout block] [dbg]
#define WGZS_DEBUG 1
#include "device_env_hip.h"
#include "kernels/common.h"
#include "kernels/conv_bf16x3.h"
#include "kernels/conv_wgrad_tr.h"
#include "kernels/conv_wgrad_zs.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NCB, int NW>
int run(int cin, int cout, int R, int rounds, int dbg) {
  typedef icl::WgTrT<NCB> C;
  typedef icl::WgZsT<NCB, NW> Z;
  const int N = 2, D = R, H = R, W = R;
  const long S = (long)D * H * W;
  std::vector<float> hx((size_t)N * cin * S), hg((size_t)N * cout * S);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f + ((s >> 24) / 256.0f) * 1e-3f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hg) v = rnd() * 0.01f;
  const int CinP = (cin + 15) / 16 * 16, CoutP = (cout + 15) / 16 * 16;
  const int pairs = ((CoutP + 16 * NCB - 1) / (16 * NCB)) * (CinP / 16);
  auto geom = [&](int tz, int ty, int tx) {
    icl::Bf3WGeom g{};
    g.Cin = cin; g.Cout = cout; g.CinP = CinP; g.CoutP = CoutP; g.D = D; g.H = H; g.W = W;
    g.ntz = (D + tz - 1) / tz; g.nty = (H + ty - 1) / ty; g.ntx = (W + tx - 1) / tx; g.ntiles = N * g.ntz * g.nty * g.ntx;
    int nsplit = 256 / pairs > 0 ? 256 / pairs : 1;
    if (nsplit > g.ntiles) nsplit = g.ntiles;
    g.tiles_per_wg = (g.ntiles + nsplit - 1) / nsplit;
    g.x_bstride = cin * S; g.gy_bstride = cout * S; g.dbg = 0;
    return g;
  };
  icl::Bf3WGeom g0 = geom(C::TZ, C::TY, C::TX), g1 = geom(1, Z::TY, Z::TX);
  g1.dbg = dbg;
  const int ns0 = (g0.ntiles + g0.tiles_per_wg - 1) / g0.tiles_per_wg, ns1 = (g1.ntiles + g1.tiles_per_wg - 1) / g1.tiles_per_wg;
  const long pe = 27L * CinP * CoutP;
  float *dx, *dg, *s0, *s1;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dg, hg.size() * 4));
  CK(hipMalloc(&s0, (size_t)ns0 * pe * 4)); CK(hipMalloc(&s1, (size_t)ns1 * pe * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_wgrad_tr_kernel<NCB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_wgrad_zs_kernel<NCB, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto go_old = [&]() { hipLaunchKernelGGL((icl::conv3d_wgrad_tr_kernel<NCB>), dim3(ns0, pairs), dim3(512), C::LDS_BYTES, 0, dx, dg, s0, g0); };
  auto go_new = [&]() { hipLaunchKernelGGL((icl::conv3d_wgrad_zs_kernel<NCB, NW>), dim3(ns1, pairs), dim3(64 * NW), Z::LDS_BYTES, 0, dx, dg, s1, g1); };
  CK(hipMemset(s0, 0xff, (size_t)ns0 * pe * 4)); CK(hipMemset(s1, 0xee, (size_t)ns1 * pe * 4));
  go_old(); go_new();
  CK(hipDeviceSynchronize()); CK(hipGetLastError());
  std::vector<float> h0((size_t)ns0 * pe), h1((size_t)ns1 * pe), h2(h1.size());
  CK(hipMemcpy(h0.data(), s0, h0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), s1, h1.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> w0(pe, 0.0), w1(pe, 0.0);
  for (int k = 0; k < ns0; ++k) for (long i = 0; i < pe; ++i) w0[i] += h0[(size_t)k * pe + i];
  for (int k = 0; k < ns1; ++k) for (long i = 0; i < pe; ++i) w1[i] += h1[(size_t)k * pe + i];
  double mx = 0, md = 0; long bad = 0;
  for (long i = 0; i < pe; ++i) { mx = std::max(mx, std::fabs(w0[i])); md = std::max(md, std::fabs(w0[i] - w1[i])); if (!(std::fabs(w1[i]) < 1e30)) ++bad; }
  int fail = (md > 2e-5 * mx || bad) ? 1 : 0;
  printf("[%d waves] %d->%d @%d^3 n=%d NCB=%d: shipped %d workgroups x %d tiles (2x4x16), z-columns %d x %d tiles (1x8x16), LDS %zu B: max |dW| %.4g, max |difference| %.3g (%.2g relative)%s\n",
         NW, cin, cout, R, N, NCB, ns0 * pairs, g0.tiles_per_wg, ns1 * pairs, g1.tiles_per_wg, Z::LDS_BYTES, mx, md, md / mx, fail ? "  MISMATCH" : "");
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipMemset(s1, 0xee, (size_t)ns1 * pe * 4));
    go_new();
    CK(hipMemcpy(h2.data(), s1, h2.size() * 4, hipMemcpyDeviceToHost));
    if (memcmp(h2.data(), h1.data(), h1.size() * 4)) { printf("  RACE: repeat %d differs from the first run\n", rep); fail = 1; }
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
  printf("  shipped <%d>: median %.1f us (min %.1f) %.1f TF | z-columns: median %.1f us (min %.1f) %.1f TF = %.3fx\n", NCB, a[a.size() / 2], a[0],
         fl / a[a.size() / 2] * 1e-6, b[b.size() / 2], b[0], fl / b[b.size() / 2] * 1e-6, a[a.size() / 2] / b[b.size() / 2]);
  g1.dbg = 0;
  go_new();
  CK(hipDeviceSynchronize());
  long long st[96];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(icl::g_wgzs_stamps), sizeof(st)));
  if (NW == 12) {
    for (int ph = 1; ph < 3; ++ph) {
      const long long *q0 = st + ph * 8, *q4 = st + (4 + ph) * 8, *q8 = st + (8 + ph) * 8;
      printf("  stamps phase %d: wave 0 multiply %lld, barrier %lld | wave 4 multiply %lld, barrier %lld | stager wave 8: split + store %lld, issue loads %lld, barrier %lld | phase %lld\n", 4 + ph,
             q0[3] - q0[0], q0[4] - q0[3], q4[3] - q4[0], q4[4] - q4[3], q8[1] - q8[0], q8[2] - q8[1], q8[4] - q8[2], q8[4] - q8[0]);
    }
  } else
  for (int w = 0; w < 2; ++w)
    for (int ph = 1; ph < 3; ++ph) {
      const long long* q = st + (w * 4 + ph) * 8;
      if (w == 0) printf("  stamps wave 0 (S -> M) phase %d: split + store %lld | issue loads %lld | multiply %lld | barrier %lld | phase %lld\n", 4 + ph, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[4] - q[0]);
      else printf("  stamps wave 4 (M -> S) phase %d: multiply %lld | split + store %lld | issue loads %lld | barrier %lld | phase %lld\n", 4 + ph, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[4] - q[0]);
    }
  for (int d : {1, 2, 3, 4, 7}) {
    g1.dbg = d;
    std::vector<double> c;
    for (int r = 0; r < 3; ++r) c.push_back(timed(go_new));
    std::sort(c.begin(), c.end());
    printf("  ablation dbg=%d (%s%s%s): %.1f us\n", d, d & 1 ? "no global loads " : "", d & 2 ? "no split/stores " : "", d & 4 ? "no multiply" : "", c[1]);
  }
  return fail;
}

int main(int argc, char** argv) {
  const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16, R = argc > 3 ? atoi(argv[3]) : 96;
  const int rounds = argc > 4 ? atoi(argv[4]) : 5;
  const int coutP = (cout + 15) / 16 * 16;
  const int force1 = argc > 5 ? atoi(argv[5]) : 0;      // 1: one cout block per workgroup column also for 32-multiples
  const int dbg = argc > 6 ? atoi(argv[6]) : 0;
  const int nw = argc > 7 ? atoi(argv[7]) : 8;
  if (coutP % 48 == 0 && coutP % 32 != 0 && !force1) return run<3, 8>(cin, cout, R, rounds, dbg);
  if (coutP % 32 == 0 && !force1) return run<2, 8>(cin, cout, R, rounds, dbg);
  return nw == 12 ? run<1, 12>(cin, cout, R, rounds, dbg) : nw == 10 ? run<1, 10>(cin, cout, R, rounds, dbg) : run<1, 8>(cin, cout, R, rounds, dbg);
}
