// Stage (a) of the blocked channels-last conversion: the loader-wave forward kernel on CL16 activations
// (csrc/kernels/conv_cl16.h) against the shipped loader-wave kernel on NCDHW (csrc/kernels/conv_bf16x3_ws.h), same weights, same values.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icl_amd/csrc -I tools/probe tools/probe/conv_cl16_probe.hip -o /tmp/conv_cl16_probe
//   /tmp/conv_cl16_probe 16 16 96 [rounds] [norm 0|1]
// norm = 1: the CL16 kernel reads RAW values and applies max(fma(v, scale, shift), 0) on load; the NCDHW kernel gets the tensor the host
// normalised with the same formula — outputs must be bit-identical either way.  Cin > 16: the input is a concatenation of two sources
// (16 channels + the rest).  Prints bitwise agreement, the epilogue statistics against a double-precision host sum, median / min launch
// times of interleaved rounds and the in-kernel stamps of three work items.
#define CL_STAMPS 1
// -DCL_DBG=n: timing ablations of the CL16 kernel (results wrong): 1 no halo loads after the first item, 2 no split / transform VALU,
// 4 every tile fetches one of eight tiles (cache-resident fetch), 8 odd workgroups start ~8 k cycles late
#include "device_env_hip.h"
#include "kernels/common.h"
#include "kernels/conv_bf16x3.h"
#include "kernels/conv_bf16x3_ws.h"
#include "probe_kernels/conv_cl16.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16, R = argc > 3 ? atoi(argv[3]) : 96;
  const int rounds = argc > 4 ? atoi(argv[4]) : 7, norm = argc > 5 ? atoi(argv[5]) : 1;
  const int N = 2, D = R, H = R, W = R;
  const long S = (long)D * H * W;
  if (cout != 16 || cin % 16) { printf("this probe covers one cout block\n"); return 2; }
  std::vector<float> hraw((size_t)N * cin * S), hx(hraw.size()), hw((size_t)cout * cin * 27), hb(cout), hss((size_t)N * cin * 2);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f + ((s >> 24) / 256.0f) * 1e-3f; };
  for (auto& v : hraw) v = rnd() * 3.f + 0.5f;
  for (auto& v : hw) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  for (int i = 0; i < N * cin; ++i) { hss[2 * i] = 0.3f + 0.05f * (i % 7); hss[2 * i + 1] = -0.2f + 0.03f * (i % 5); }
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < cin; ++c)
      for (long p = 0; p < S; ++p) {
        const size_t i = ((size_t)n * cin + c) * S + p;
        hx[i] = norm ? fmaxf(fmaf(hraw[i], hss[2 * (n * cin + c)], hss[2 * (n * cin + c) + 1]), 0.f) : hraw[i];
      }
  // weights: fp32 pack wp[tap][cin][coutP], split on the device
  const int coutP = 16, cinP = cin;
  const long wpn = (long)27 * cinP * coutP;
  std::vector<float> hwp(wpn, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < 27; ++t) hwp[((size_t)t * cinP + ci) * coutP + co] = hw[((size_t)co * cin + ci) * 27 + t];
  float *dx, *draw, *db, *dy0, *dy1, *dwp, *dss0, *dss1, *dstats, *dssout;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&draw, hx.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dy0, (size_t)N * cout * S * 4)); CK(hipMalloc(&dy1, (size_t)N * cout * S * 4));
  CK(hipMalloc(&dwp, wpn * 4)); CK(hipMemcpy(dwp, hwp.data(), wpn * 4, hipMemcpyHostToDevice));
  uint4* dws; const long items = (long)(cin / 16) * 3 * 2 * icl::Bf3::SLOTS * coutP; CK(hipMalloc(&dws, items * 3 * 16));
  hipLaunchKernelGGL(icl::conv_bf16x3_split_weights_kernel, dim3(64), dim3(256), 0, 0, dwp, dws, cinP, coutP, cin / 16);
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  // the raw tensor as (up to) two CL16 sources: channels [0, 16) and [16, cin)
  const int c0 = 16, c1 = cin - 16;
  std::vector<float> hs0((size_t)N * c0 * S), hs1((size_t)N * (c1 > 0 ? c1 : 1) * S), hss0((size_t)N * c0 * 2), hss1((size_t)N * (c1 > 0 ? c1 : 1) * 2);
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < cin; ++c) {
      std::vector<float>& dst = c < c0 ? hs0 : hs1;
      std::vector<float>& dss = c < c0 ? hss0 : hss1;
      const int cc = c < c0 ? c : c - c0, C = c < c0 ? c0 : c1;
      for (long p = 0; p < S; ++p) dst[(((size_t)n * (C / 16) + cc / 16) * S + p) * 16 + (cc & 15)] = hraw[((size_t)n * cin + c) * S + p];
      dss[((size_t)n * C + cc) * 2] = hss[2 * (n * cin + c)];
      dss[((size_t)n * C + cc) * 2 + 1] = hss[2 * (n * cin + c) + 1];
    }
  float *ds0, *ds1;
  CK(hipMalloc(&ds0, hs0.size() * 4)); CK(hipMalloc(&ds1, hs1.size() * 4)); CK(hipMalloc(&dss0, hss0.size() * 4)); CK(hipMalloc(&dss1, hss1.size() * 4));
  CK(hipMemcpy(ds0, hs0.data(), hs0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds1, hs1.data(), hs1.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dss0, hss0.data(), hss0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dss1, hss1.data(), hss1.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy0, 0xff, (size_t)N * cout * S * 4)); CK(hipMemset(dy1, 0xee, (size_t)N * cout * S * 4));

  icl::Bf3Geom g{};
  g.Cin = cin; g.Cout = cout; g.CinP = cinP; g.CoutP = coutP; g.D = D; g.H = H; g.W = W;
  g.ntz = (D + 3) / 4; g.nty = (H + 7) / 8; g.ntx = (W + 15) / 16;
  g.ntiles = N * g.ntz * g.nty * g.ntx;
  g.nchunks = cin / 16; g.x_bstride = cin * S; g.y_bstride = cout * S;
  const int gx = g.ntiles < 256 ? (g.ntiles + 7) / 8 * 8 : 256;
  CK(hipMalloc(&dstats, (size_t)N * cout * gx * 3 * 4)); CK(hipMalloc(&dssout, (size_t)N * cout * 2 * 4));
  float* dstats0; CK(hipMalloc(&dstats0, (size_t)N * cout * gx * 3 * 4));
  g.stats = dstats0; g.nbatch = N; g.wgs = gx;      // both kernels produce the InstanceNorm statistics of their output
  icl::ClGeom c{};
  c.src[0] = icl::ClSrc{ds0, (long)c0 * S, norm ? dss0 : nullptr, c0 / 16};
  c.src[1] = icl::ClSrc{ds1, (long)c1 * S, norm ? dss1 : nullptr, c1 / 16};
  c.Cout = cout; c.CoutP = coutP; c.D = D; c.H = H; c.W = W; c.ntz = g.ntz; c.nty = g.nty; c.ntx = g.ntx; c.ntiles = g.ntiles;
  c.nchunks = cin / 16; c.y_bstride = cout * S; c.stats = dstats; c.nbatch = N;
  const size_t lds = icl::Bf3T<8>::lds_bytes(1, 3);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_ws_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_cl16_fwd_ws_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  bool with_stats = true;
  auto go_old = [&]() { hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_ws_kernel<1, false>), dim3(gx, 1), dim3(768), lds, 0, dx, dws, db, dy0, g); };
  auto go_new = [&]() {
    icl::ClGeom cc = c;
    if (!with_stats) cc.stats = nullptr;
    hipLaunchKernelGGL((icl::conv3d_cl16_fwd_ws_kernel<1>), dim3(gx, 1), dim3(768), lds, 0, dws, db, dy1, cc);
  };
  go_old(); go_new();
  hipLaunchKernelGGL(icl::cl_stats_finalize_kernel, dim3((N * cout + 63) / 64), dim3(64), 0, 0, dstats, N * cout, gx, 1e-5f, (float*)nullptr, (float*)nullptr, dssout);
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  std::vector<float> h0((size_t)N * cout * S), h1(h0.size()), hso((size_t)N * cout * 2);
  CK(hipMemcpy(h0.data(), dy0, h0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), dy1, h1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hso.data(), dssout, hso.size() * 4, hipMemcpyDeviceToHost));
  size_t ndiff = 0; double maxd = 0;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < cout; ++co)
      for (long p = 0; p < S; ++p) {
        const float a = h0[((size_t)n * cout + co) * S + p], b = h1[(((size_t)n * (cout / 16) + co / 16) * S + p) * 16 + (co & 15)];
        if (memcmp(&a, &b, 4)) { if (ndiff++ < 5) printf("  diff at n %d co %d p %ld: shipped %g cl16 %g\n", n, co, p, a, b); }
        const double d = fabs((double)a - b); if (d > maxd || d != d) maxd = d;
      }
#if defined(CL_DBG)
  printf("ABLATION BUILD CL_DBG=%d (outputs are expected to differ)\n", CL_DBG);
#endif
  printf("%d->%d @%d^3 n=%d norm-on-load %d: %zu of %zu outputs differ bitwise (max |diff| %.3e)\n", cin, cout, R, N, norm, ndiff, h0.size(), maxd);
  double worst_scale = 0, worst_shift = 0;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < cout; ++co) {
      double s1 = 0, s2 = 0;
      const float* yv = h0.data() + ((size_t)n * cout + co) * S;
      for (long p = 0; p < S; ++p) s1 += yv[p];
      const double mean = s1 / S;
      for (long p = 0; p < S; ++p) s2 += (yv[p] - mean) * (yv[p] - mean);
      const double rstd = 1.0 / sqrt(s2 / S + 1e-5);
      worst_scale = std::max(worst_scale, fabs(hso[2 * (n * cout + co)] - rstd) / rstd);
      worst_shift = std::max(worst_shift, fabs(hso[2 * (n * cout + co) + 1] + mean * rstd) / (fabs(mean * rstd) + 1.0));
    }
  printf("  epilogue statistics -> (scale, shift): worst relative error of scale %.2e, of shift %.2e (against double-precision host sums)\n", worst_scale, worst_shift);
  for (int rep = 0; rep < 6; ++rep) {      // race screen
    CK(hipMemset(dy1, 0xee, (size_t)N * cout * S * 4));
    go_new();
    std::vector<float> h2(h0.size());
    CK(hipMemcpy(h2.data(), dy1, h2.size() * 4, hipMemcpyDeviceToHost));
    if (memcmp(h2.data(), h1.data(), h1.size() * 4)) { printf("  RACE: repeat %d differs from the first run\n", rep); ndiff++; }
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](auto&& fn) {
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
  };
  std::vector<double> t_old, t_new, t_nos;
  for (int r = 0; r <= rounds; ++r) {
    with_stats = true;
    const double a = timed(go_old), b = timed(go_new);
    with_stats = false;
    const double cns = timed(go_new);
    if (r) { t_old.push_back(a); t_new.push_back(b); t_nos.push_back(cns); }
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mn = [](const std::vector<double>& v) { return *std::min_element(v.begin(), v.end()); };
  const double fl = 2.0 * 27 * cin * cout * N * S;
  printf("  shipped ws<1> NCDHW: median %.1f us (min %.1f) %.1f TF | CL16 ws<1> with statistics: median %.1f us (min %.1f) %.1f TF = %.3fx | without statistics %.1f us (min %.1f)\n",
         med(t_old), mn(t_old), fl / med(t_old) * 1e-6, med(t_new), mn(t_new), fl / med(t_new) * 1e-6, med(t_old) / med(t_new), med(t_nos), mn(t_nos));
  with_stats = true;
  go_new();
  CK(hipDeviceSynchronize());
  long long st[144];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(icl::g_cl_stamps), sizeof(st)));
  for (int w = 0; w < 2; ++w)
    for (int it = 0; it < 3; ++it) {
      const long long* q = st + (w * 3 + it) * 16;
      printf("  cl16 stamps consumer wave %d item %d: wait-A %lld | dz0 %lld dz1 %lld dz2 %lld | epilogue %lld | wait-B %lld | item total %lld\n",
             4 * w, it + 2, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[6] - q[0]);
    }
  for (int it = 0; it < 3; ++it) {
    const long long* q = st + (2 * 3 + it) * 16;
    printf("  cl16 stamps loader wave 8 item %d: wait-A %lld | issue %lld first-round %lld rest %lld | wait-B %lld | deposit %lld | item total %lld\n",
           it + 2, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[6] - q[0]);
  }
  return ndiff ? 1 : 0;
}
