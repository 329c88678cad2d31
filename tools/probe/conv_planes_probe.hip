// Standalone A/B of csrc/kernels/conv_planes.h (halo tiles by LDS-DMA from producer-written operand planes) against the shipped
// conv3d_bf16x3_fwd_kernel<NBT, 8, 60> (fp32 input, split while staging):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icl_amd/csrc -I tools/probe tools/probe/conv_planes_probe.hip -o tools/probe/planesprobe
//   tools/probe/planesprobe 16 16 96        (cin cout side [rounds])
// Prints whether the outputs are bit-identical and median / min launch times of interleaved rounds; first a micro-test of what the
// hardware writes to LDS for a range-checked LDS-DMA lane (information only: the kernel does not rely on it).
#define PLANES_STAMPS 1
#define WS_STAMPS 1
#include "device_env_hip.h"
#include "kernels/common.h"
#include "kernels/conv_bf16x3.h"
#include "probe_kernels/conv_planes.h"
#include "kernels/conv_bf16x3_ws.h"
#include "probe_kernels/conv_bf16x3_wsr.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void oob_probe_kernel(const uint4* g, uint4* out, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  uint4* lds = (uint4*)raw;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 128; i += blockDim.x) lds[i] = make_uint4(0xdeadbeefu, 1u, 2u, 3u);
  __syncthreads();
  icl_rsrc_t r = icl_make_rsrc(g, (unsigned)n * 16u);
  const unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)lane * 16u;
  icl_buffer_load_lds_b128(r, lds, voff, 0u);
  ICL_WAIT_VMEM();
  __syncthreads();
  for (int i = threadIdx.x; i < 128; i += blockDim.x) out[i] = lds[i];
}

template <int NBT>
void launch_old(const float* x, const uint4* ws, const float* bias, float* y, icl::Bf3Geom g, bool flat) {
  const int gy = (g.CoutP + 16 * NBT - 1) / (16 * NBT);
  const int gx = g.ntiles < 256 ? (g.ntiles + 7) / 8 * 8 : 256;
  if (flat) {
    const size_t lds = icl::Bf3F24::lds_bytes(NBT, NBT == 1 ? 3 : 1);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_kernel<NBT, 8, 60, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_kernel<NBT, 8, 60, true>), dim3(gx, gy), dim3(512), lds, 0, x, ws, bias, y, g);
  } else {
    const size_t lds = icl::Bf3T<8>::lds_bytes(NBT, NBT == 1 ? 3 : 1);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_kernel<NBT, 8, 60>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_kernel<NBT, 8, 60>), dim3(gx, gy), dim3(512), lds, 0, x, ws, bias, y, g);
  }
}
template <int NBT, int TY, int NWV, bool DBUF, bool FLAT>
void launch_cfg(const uint4* planes, const uint4* ws, const float* bias, float* y, icl::Bf3PGeom g, int maxw) {
  typedef icl::PlanesCfg<NBT, TY, NWV, DBUF, FLAT> C;
  const int gy = (g.CoutP + 16 * NBT - 1) / (16 * NBT);
  if (!FLAT) { g.nty = (g.H + TY - 1) / TY; g.ntiles = 2 * g.ntz * g.nty * g.ntx; }
  const int gx = g.ntiles < maxw ? (g.ntiles + 7) / 8 * 8 : maxw;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_planes_fwd_kernel<NBT, TY, NWV, DBUF, FLAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((icl::conv3d_planes_fwd_kernel<NBT, TY, NWV, DBUF, FLAT>), dim3(gx, gy), dim3(64 * NWV), C::LDS_BYTES, 0, planes, ws, bias, y, g);
}
// cfg 3: conv_bf16x3_ws.h — fp32 input, eight consumer waves + four loader waves
template <int NBT>
void launch_ws(const float* x, const uint4* ws, const float* bias, float* y, icl::Bf3Geom g, bool flat) {
  const int gy = (g.CoutP + 16 * NBT - 1) / (16 * NBT);
  const int gx = g.ntiles < 256 ? (g.ntiles + 7) / 8 * 8 : 256;
  if (flat) {
    const size_t lds = icl::Bf3F24::lds_bytes(NBT, NBT == 1 ? 3 : 1);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_ws_kernel<NBT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_ws_kernel<NBT, true>), dim3(gx, gy), dim3(768), lds, 0, x, ws, bias, y, g);
  } else {
    const size_t lds = icl::Bf3T<8>::lds_bytes(NBT, NBT == 1 ? 3 : 1);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_ws_kernel<NBT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_ws_kernel<NBT, false>), dim3(gx, gy), dim3(768), lds, 0, x, ws, bias, y, g);
  }
}
// cfg 4: conv3d_bf16x3_fwd_wsr_kernel — loader waves + NCBLK cout blocks per workgroup, weight planes through a three-slot DMA ring
template <int NBT>
void launch_wsr(const float* x, const uint4* ws, const float* bias, float* y, icl::Bf3Geom g, bool flat) {
  if constexpr (NBT >= 2) {
    const int gy = (g.CoutP + 16 * NBT - 1) / (16 * NBT);
    const int gx = g.ntiles < 256 ? (g.ntiles + 7) / 8 * 8 : 256;
    if (flat) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_wsr_kernel<NBT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_wsr_kernel<NBT, true>), dim3(gx, gy), dim3(768), icl::Bf3F24::lds_bytes(1, 3), 0, x, ws, bias, y, g);
    } else {
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&icl::conv3d_bf16x3_fwd_wsr_kernel<NBT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      hipLaunchKernelGGL((icl::conv3d_bf16x3_fwd_wsr_kernel<NBT, false>), dim3(gx, gy), dim3(768), icl::Bf3T<8>::lds_bytes(1, 3), 0, x, ws, bias, y, g);
    }
  }
}
// cfg 0: <NBT, 8, 8, single> one workgroup per CU; 1: <1, 4, 4, single> two workgroups per CU; 2: <NBT, 4, 8, double-buffered>
template <int NBT>
void launch_new(const uint4* planes, const uint4* ws, const float* bias, float* y, icl::Bf3PGeom g, bool flat, int cfg) {
  if (flat) { launch_cfg<NBT, 8, 8, false, true>(planes, ws, bias, y, g, 256); return; }
  if (cfg == 1) { if constexpr (NBT == 1) launch_cfg<1, 4, 4, false, false>(planes, ws, bias, y, g, 512); return; }
  if (cfg == 2) { if constexpr (NBT <= 2) launch_cfg<NBT, 4, 8, true, false>(planes, ws, bias, y, g, 256); return; }
  launch_cfg<NBT, 8, 8, false, false>(planes, ws, bias, y, g, 256);
}

int main(int argc, char** argv) {
  const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16, R = argc > 3 ? atoi(argv[3]) : 96;
  const int rounds = argc > 4 ? atoi(argv[4]) : 7, cfg = argc > 5 ? atoi(argv[5]) : 0;
  const int nbt = cout % 48 == 0 ? 3 : cout % 32 == 0 ? 2 : 1, N = 2;
  const int D = R, H = R, W = R;
  const bool flat = W == 24;
  const long S = (long)D * H * W;
  if (getenv("PROBE_OOB")) {
    std::vector<uint4> hg(64);
    for (int i = 0; i < 64; ++i) hg[i] = make_uint4(100u + i, 0u, 0u, 0u);
    uint4 *dg, *dout;
    CK(hipMalloc(&dg, 64 * 16)); CK(hipMalloc(&dout, 128 * 16));
    CK(hipMemcpy(dg, hg.data(), 64 * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(oob_probe_kernel, dim3(1), dim3(64), 128 * 16, 0, dg, dout, 64);
    std::vector<uint4> ho(128);
    CK(hipMemcpy(ho.data(), dout, 128 * 16, hipMemcpyDeviceToHost));
    printf("LDS-DMA range check: in-range lane 0 -> %u (expect 100), lane 2 -> %u (102); range-checked lane 1 -> 0x%x, lane 3 -> 0x%x (0 = zero written, 0xdeadbeef = LDS untouched); slot 64 (never addressed) 0x%x\n",
           ho[0].x, ho[2].x, ho[1].x, ho[3].x, ho[64].x);
  }
  std::vector<float> hx((size_t)N * cin * S), hw((size_t)cout * cin * 27), hb(cout);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f + ((s >> 24) / 256.0f) * 1e-3f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  const int coutP = (cout + 15) / 16 * 16, cinP = cin;
  const long wpn = (long)27 * cinP * coutP;
  std::vector<float> hwp(wpn, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < 27; ++t) hwp[((size_t)t * cinP + ci) * coutP + co] = hw[((size_t)co * cin + ci) * 27 + t];
  float *dx, *db, *dy0, *dy1, *dwp;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dy0, (size_t)N * cout * S * 4)); CK(hipMalloc(&dy1, (size_t)N * cout * S * 4));
  CK(hipMalloc(&dwp, wpn * 4)); CK(hipMemcpy(dwp, hwp.data(), wpn * 4, hipMemcpyHostToDevice));
  uint4* dws; const long items = (long)(cin / 16) * 3 * 2 * icl::Bf3::SLOTS * coutP; CK(hipMalloc(&dws, items * 3 * 16));
  hipLaunchKernelGGL(icl::conv_bf16x3_split_weights_kernel, dim3(64), dim3(256), 0, 0, dwp, dws, cinP, coutP, cin / 16);
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy0, 0xff, (size_t)N * cout * S * 4)); CK(hipMemset(dy1, 0xee, (size_t)N * cout * S * 4));
  const long pitch = icl::planes_pitch(S), pbs = (long)(cin / 16) * 6 * pitch;
  uint4* dpl; CK(hipMalloc(&dpl, (size_t)N * pbs * 16));
  CK(hipMemset(dpl, 0x7f, (size_t)N * pbs * 16));

  icl::Bf3Geom g{};
  g.Cin = cin; g.Cout = cout; g.CinP = cinP; g.CoutP = coutP; g.D = D; g.H = H; g.W = W;
  if (flat) { g.ntz = (D + 1) / 2; g.nty = (H + 7) / 8; g.ntx = 1; }
  else { g.ntz = (D + 3) / 4; g.nty = (H + 7) / 8; g.ntx = (W + 15) / 16; }
  g.ntiles = N * g.ntz * g.nty * g.ntx;
  g.nchunks = cin / 16; g.x_bstride = cin * S; g.y_bstride = cout * S;
  icl::Bf3PGeom p{};
  p.Cout = cout; p.CoutP = coutP; p.D = D; p.H = H; p.W = W; p.ntz = g.ntz; p.nty = g.nty; p.ntx = g.ntx; p.ntiles = g.ntiles;
  p.nchunks = g.nchunks; p.ppitch = pitch; p.p_bstride = pbs; p.y_bstride = g.y_bstride;
  p.dbg = 0;

  auto split = [&]() { hipLaunchKernelGGL(icl::planes_from_f32_kernel, dim3(2048), dim3(256), 0, 0, dx, dpl, N, cin, S, (long)cin * S, pbs); };
  auto go_old = [&]() { if (nbt == 1) launch_old<1>(dx, dws, db, dy0, g, flat); else if (nbt == 2) launch_old<2>(dx, dws, db, dy0, g, flat); else launch_old<3>(dx, dws, db, dy0, g, flat); };
  auto go_ws = [&]() { if (nbt == 1) launch_ws<1>(dx, dws, db, dy1, g, flat); else if (nbt == 2) launch_ws<2>(dx, dws, db, dy1, g, flat); else launch_ws<3>(dx, dws, db, dy1, g, flat); };
  // cfg 10 + f: the shipped kernel with Bf3Geom::flags = f (1 y-slowest tile order, 2 non-temporal output stores)
  auto go_flags = [&]() { icl::Bf3Geom gf = g; gf.flags = cfg - 10; if (nbt == 1) launch_old<1>(dx, dws, db, dy1, gf, flat); else if (nbt == 2) launch_old<2>(dx, dws, db, dy1, gf, flat); else launch_old<3>(dx, dws, db, dy1, gf, flat); };
  auto go_wsr = [&]() { if (nbt == 2) launch_wsr<2>(dx, dws, db, dy1, g, flat); else if (nbt == 3) launch_wsr<3>(dx, dws, db, dy1, g, flat); };
  auto go_new = [&]() { if (cfg >= 10) { go_flags(); return; } if (cfg == 4) { go_wsr(); return; } if (cfg == 3) { go_ws(); return; } if (nbt == 1) launch_new<1>(dpl, dws, db, dy1, p, flat, cfg); else if (nbt == 2) launch_new<2>(dpl, dws, db, dy1, p, flat, cfg); else launch_new<3>(dpl, dws, db, dy1, p, flat, cfg); };
  split(); go_old(); go_new();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  std::vector<float> h0((size_t)N * cout * S), h1(h0.size());
  CK(hipMemcpy(h0.data(), dy0, h0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), dy1, h1.size() * 4, hipMemcpyDeviceToHost));
  size_t ndiff = 0; double maxd = 0;
  for (size_t i = 0; i < h0.size(); ++i) {
    if (memcmp(&h0[i], &h1[i], 4)) { if (ndiff++ < 5) printf("  diff at %zu: shipped %g planes %g\n", i, h0[i], h1[i]); }
    const double d = fabs((double)h0[i] - h1[i]); if (d > maxd || d != d) maxd = d;
  }
  printf("%d->%d @%d^3 n=%d nbt=%d%s cfg %d: %zu of %zu outputs differ bitwise (max |diff| %.3e)\n", cin, cout, R, N, nbt, flat ? " flat24" : "", cfg, ndiff, h0.size(), maxd);
  // race screen: repeat the planes kernel and compare with its first result
  for (int rep = 0; rep < 10; ++rep) {
    CK(hipMemset(dy1, 0xee, (size_t)N * cout * S * 4));
    go_new();
    std::vector<float> h2(h0.size());
    CK(hipMemcpy(h2.data(), dy1, h2.size() * 4, hipMemcpyDeviceToHost));
    if (memcmp(h2.data(), h1.data(), h1.size() * 4)) { printf("  RACE: repeat %d of the planes kernel differs from its first run\n", rep); ndiff++; }
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
  std::vector<double> t_old, t_new, t_split;
  for (int r = 0; r <= rounds; ++r) {
    const double a = timed(go_old), b = timed(go_new), c = timed(split);
    if (r) { t_old.push_back(a); t_new.push_back(b); t_split.push_back(c); }
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mn = [](const std::vector<double>& v) { return *std::min_element(v.begin(), v.end()); };
  const double fl = 2.0 * 27 * cin * cout * N * S;
  printf("  shipped <%d,8,60>: median %.1f us (min %.1f) %.1f TF | planes + LDS-DMA: median %.1f us (min %.1f) %.1f TF = %.3fx | stand-alone split kernel %.1f us\n",
         nbt, med(t_old), mn(t_old), fl / med(t_old) * 1e-6, med(t_new), mn(t_new), fl / med(t_new) * 1e-6, med(t_old) / med(t_new), med(t_split));
  if (cfg >= 10 || cfg == 4) return ndiff ? 1 : 0;
  if (cfg == 3) {
    long long st[144];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(icl::g_ws_stamps), sizeof(st)));
    for (int w = 0; w < 2; ++w)
      for (int it = 0; it < 3; ++it) {
        const long long* q = st + (w * 3 + it) * 16;
        printf("  ws stamps consumer wave %d item %d: wait-A %lld | dz0 %lld dz1 %lld dz2 %lld | epilogue %lld | wait-B %lld | item total %lld\n",
               4 * w, it + 2, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[6] - q[0]);
      }
    for (int it = 0; it < 3; ++it) {
      const long long* q = st + (2 * 3 + it) * 16;
      printf("  ws stamps loader wave 8 item %d: wait-A %lld | phase0 %lld phase1 %lld phase2+split %lld | wait-B %lld | deposit %lld | item total %lld\n",
             it + 2, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[6] - q[0]);
    }
    return ndiff ? 1 : 0;
  }
  {
    long long st[96];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(icl::g_planes_stamps), sizeof(st)));
    for (int w = 0; w < 2; ++w)
      for (int it = 0; it < 3; ++it) {
        const long long* q = st + (w * 3 + it) * 16;
        printf("  stamps wave %s item %d: dma-wait %lld | barrier %lld | dz0 %lld dz1 %lld dz2 %lld | end-barrier %lld issue %lld epilogue %lld | item total %lld\n",
               w ? "N/2" : "0", it + 2, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[7] - q[6], q[8] - q[7], q[8] - q[0]);
      }
  }
  for (int dbg = 1; dbg <= 3; ++dbg) {
    p.dbg = dbg;
    std::vector<double> t;
    for (int r = 0; r < 4; ++r) { const double a = timed(go_new); if (r) t.push_back(a); }
    printf("  ablation dbg=%d (%s): median %.1f us\n", dbg, dbg == 1 ? "no halo DMA after the first item" : dbg == 2 ? "no output stores" : "neither", med(t));
  }
  return ndiff ? 1 : 0;
}
