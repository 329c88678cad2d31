// What does it cost a CU to pull the halo tile of the split-product convolution, by activation layout and load shape?  (round 4)
// Every workgroup (512 threads, one per CU) walks the 4 x 8 x 16 tiles of a 16-channel 96^3 x 2 tensor in the forward kernel's XCD-local
// order and fetches the 6 x 10 x 18 halo of each into registers (summed into a checksum: nothing else runs).  Variants:
//   0  NCDHW fp32, one-dword gathers: a lane reads one halo position of 8 channel planes (the shipped kernel's staging loads)
//   1  NCDHW fp32, aligned x-quads: a lane reads 16 bytes (4 consecutive x of one channel), 6 quads per row of 18 (33 % over-fetch)
//   2  channels-last fp32 [voxel][16 ch]: a lane reads 2 x 16 bytes = 8 channels of one position
//   3  bf16 planes [q 6][voxel] of 16 bytes (what conv_planes.h fetches): a lane reads 16 bytes
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icl_amd/csrc tools/probe/halo_fetch_probe.hip -o tools/probe/halofetch && tools/probe/halofetch
#include "device_env_hip.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int TZ = 4, TY = 8, TX = 16, PZ = 6, PY = 10, PX = 18, NPOS = PZ * PY * PX;

template <int V>
__global__ __launch_bounds__(512) void fetch_kernel(const float* __restrict__ x, float* __restrict__ out, int D, int H, int W, int nb) {
  const int tid = threadIdx.x;
  const int HW = H * W;
  const long DHW = (long)D * HW;
  const int ntz = D / TZ, nty = H / TY, ntx = W / TX, tiles_per = ntz * nty * ntx, ntiles = nb * tiles_per;
  const int per_xcd = (ntiles + 7) / 8, xcd = blockIdx.x & 7, wgs = gridDim.x >> 3;
  const int xcd_end = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
  float acc = 0.f;
  for (int tile = xcd * per_xcd + (blockIdx.x >> 3); tile < xcd_end; tile += wgs) {
    const int b = tile / tiles_per, bt = tile % tiles_per;
    const int x0 = (bt % ntx) * TX, y0 = ((bt / ntx) % nty) * TY, z0 = (bt / (ntx * nty)) * TZ;
    if (V == 0) {
      const icl_rsrc_t r = icl_make_rsrc(x + (long)b * 16 * DHW, (unsigned)(16 * DHW * 4));
      float v[5][8];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int it = tid + k * 512, o = it / NPOS, pos = it % NPOS;
        const int px = pos % PX, row = pos / PX, py = row % PY, pz = row / PY;
        const int gz = z0 - 1 + pz, gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = (it < 2 * NPOS) & ((unsigned)gz < (unsigned)D) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        const unsigned off = ok ? (unsigned)(o * 8 * (int)DHW + gz * HW + gy * W + gx) * 4u : 0x80000000u;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[k][c] = icl_buffer_load_f32(r, off, (unsigned)c * (unsigned)DHW * 4u);
      }
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += v[k][c];
    } else if (V == 1) {
      const icl_rsrc_t r = icl_make_rsrc(x + (long)b * 16 * DHW, (unsigned)(16 * DHW * 4));
      constexpr int ITEMS = 16 * PZ * PY * 6, R = (ITEMS + 511) / 512;      // (channel, row, quad)
      uint4 v[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int it = tid + k * 512, q = it % 6, row = (it / 6) % (PZ * PY), c = it / (6 * PZ * PY);
        const int gz = z0 - 1 + row / PY, gy = y0 - 1 + row % PY, gx = x0 - 4 + 4 * q;
        const bool ok = (it < ITEMS) & ((unsigned)gz < (unsigned)D) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        v[k] = icl_buffer_load_u32x4(r, ok ? (unsigned)(c * (int)DHW + gz * HW + gy * W + gx) * 4u : 0x80000000u);
      }
#pragma unroll
      for (int k = 0; k < R; ++k) acc += __uint_as_float(v[k].x) + __uint_as_float(v[k].y) + __uint_as_float(v[k].z) + __uint_as_float(v[k].w);
    } else if (V == 2) {
      const icl_rsrc_t r = icl_make_rsrc(x + (long)b * 16 * DHW, (unsigned)(16 * DHW * 4));
      uint4 v[5][2];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int it = tid + k * 512, o = it & 1, pos = it >> 1;      // consecutive lanes: the two octets of one position, then the next position
        const int px = pos % PX, row = pos / PX, py = row % PY, pz = row / PY;
        const int gz = z0 - 1 + pz, gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = (it < 2 * NPOS) & ((unsigned)gz < (unsigned)D) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        const unsigned off = ok ? (unsigned)((gz * HW + gy * W + gx) * 16 + o * 8) * 4u : 0x80000000u;
        v[k][0] = icl_buffer_load_u32x4(r, off);
        v[k][1] = icl_buffer_load_u32x4(r, ok ? off + 16u : 0x80000000u);
      }
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc += __uint_as_float(v[k][h].x) + __uint_as_float(v[k][h].y) + __uint_as_float(v[k][h].z) + __uint_as_float(v[k][h].w);
    } else {
      // planes: 6 x DHW x 16 bytes per sample (same byte count as 24 fp32 channels)
      const icl_rsrc_t r = icl_make_rsrc(x + (long)b * 24 * DHW, (unsigned)(24 * DHW * 4));
      constexpr int ITEMS = 6 * NPOS, R = (ITEMS + 511) / 512;
      uint4 v[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int it = tid + k * 512, q = it / NPOS, pos = it % NPOS;
        const int px = pos % PX, row = pos / PX, py = row % PY, pz = row / PY;
        const int gz = z0 - 1 + pz, gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = (it < ITEMS) & ((unsigned)gz < (unsigned)D) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        v[k] = icl_buffer_load_u32x4(r, ok ? (unsigned)(q * (int)DHW + gz * HW + gy * W + gx) * 16u : 0x80000000u);
      }
#pragma unroll
      for (int k = 0; k < R; ++k) acc += __uint_as_float(v[k].x) + __uint_as_float(v[k].y) + __uint_as_float(v[k].z) + __uint_as_float(v[k].w);
    }
  }
  if (acc == 12345.678f) out[blockIdx.x * 512 + tid] = acc;      // keeps the loads alive
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 96, nb = 2;
  const long DHW = (long)R * R * R;
  float *x, *out;
  CK(hipMalloc(&x, (size_t)nb * 24 * DHW * 4)); CK(hipMalloc(&out, 256 * 512 * 4));
  CK(hipMemset(x, 0x11, (size_t)nb * 24 * DHW * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"NCDHW one-dword gathers (shipped staging)", "NCDHW aligned x-quads, 16 B", "channels-last [voxel][16 ch], 2 x 16 B", "bf16 planes, 16 B"};
  const double useful[4] = {NPOS * 16 * 4.0, NPOS * 16 * 4.0, NPOS * 16 * 4.0, NPOS * 16 * 6.0};
  const int ntiles = nb * (R / TZ) * (R / TY) * (R / TX);
  for (int rep = 0; rep < 2; ++rep)
    for (int v = 0; v < 4; ++v) {
      auto go = [&]() {
        if (v == 0) hipLaunchKernelGGL(fetch_kernel<0>, dim3(256), dim3(512), 0, 0, x, out, R, R, R, nb);
        else if (v == 1) hipLaunchKernelGGL(fetch_kernel<1>, dim3(256), dim3(512), 0, 0, x, out, R, R, R, nb);
        else if (v == 2) hipLaunchKernelGGL(fetch_kernel<2>, dim3(256), dim3(512), 0, 0, x, out, R, R, R, nb);
        else hipLaunchKernelGGL(fetch_kernel<3>, dim3(256), dim3(512), 0, 0, x, out, R, R, R, nb);
      };
      for (int i = 0; i < 3; ++i) go();
      std::vector<double> t;
      for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) go();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 100.0);
      }
      std::sort(t.begin(), t.end());
      const double us = t[3], items = (double)ntiles / 256;
      if (rep) printf("%d^3 x %d, %d tiles (%.1f per CU): %-45s %7.1f us per launch = %5.2f us per tile and CU, %5.1f useful B/clk/CU at 2.1 GHz, %5.2f TB/s useful chip-wide\n",
                      R, nb, ntiles, items, names[v], us, us / items, useful[v] / (us / items * 2100.0), useful[v] * ntiles / us * 1e-6);
    }
  return 0;
}
