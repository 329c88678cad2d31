// datafeed.h — on-device training augmentation of the 3-D trainers: RandomRotFlip -> RandomCrop -> ToTensor
// (/root/reference/code/dataloaders/brats2019.py:80-147,177-189; composed at train_inherent_consistent_unet_3D_BraTS.py:66-73)
// as ONE gather kernel over volumes that stay resident in HBM (the whole BraTS2019 training set is 9 GB of fp32; 288 GB per GPU).
// The host draws the random parameters with the reference's numpy call sequence; the kernel maps every output voxel back through
// crop offset -> zero padding -> flip -> rot90 and reads it from the source volume.  The depth axis (fastest) is untouched by
// rot90 / flip (they act on axes 0, 1), so reads and writes are coalesced along it.
#pragma once

namespace icl {

constexpr int kFeedMaxBatch = 16;

struct FeedSample {
  const float* image;        // [n0][n1][n2] fp32
  const uint8_t* label;      // [n0][n1][n2] uint8
  int n0, n1, n2;            // source extents
  int k;                     // np.rot90(., k) on axes (0, 1), k in 0..3
  int flip_axis;             // np.flip axis: 0 or 1; -1 = none
  int p0, p1, p2;            // zero padding added on both sides of each axis (RandomCrop pads small volumes)
  int c0, c1, c2;            // crop origin in the padded, rotated, flipped volume
};

struct FeedBatch {
  FeedSample s[kFeedMaxBatch];
};

// image_out [B][1][o0][o1][o2] fp32, label_out [B][o0][o1][o2] int64.  grid (ceil(o2/64)... flat), one thread per output voxel.
__global__ __launch_bounds__(256) void crop_rotflip_kernel(FeedBatch fb, float* __restrict__ image_out, long long* __restrict__ label_out, int B,
                                                           int o0, int o1, int o2) {
  const long per = (long)o0 * o1 * o2, total = per * B;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e / per);
    long r = e % per;
    const int l = (int)(r % o2);
    r /= o2;
    const int j = (int)(r % o1), i = (int)(r / o1);
    const FeedSample& s = fb.s[b];
    // extents after rot90: odd k swaps axes 0 and 1
    const int r0 = (s.k & 1) ? s.n1 : s.n0, r1 = (s.k & 1) ? s.n0 : s.n1;
    // crop -> padded coordinates -> unpadded (rotated + flipped) coordinates
    int a = s.c0 + i - s.p0, c = s.c1 + j - s.p1;
    const int d = s.c2 + l - s.p2;
    float v = 0.f;
    long long lab = 0;
    if (a >= 0 && a < r0 && c >= 0 && c < r1 && d >= 0 && d < s.n2) {
      if (s.flip_axis == 0) a = r0 - 1 - a;
      else if (s.flip_axis == 1) c = r1 - 1 - c;
      // rot90 on axes (0, 1): rotated[a][c] = source[sa][sc]
      int sa, sc;
      switch (s.k & 3) {
        case 0: sa = a; sc = c; break;
        case 1: sa = c; sc = s.n1 - 1 - a; break;
        case 2: sa = s.n0 - 1 - a; sc = s.n1 - 1 - c; break;
        default: sa = s.n0 - 1 - c; sc = a; break;
      }
      const long src = ((long)sa * s.n1 + sc) * s.n2 + d;
      v = s.image[src];
      lab = (long long)s.label[src];
    }
    image_out[e] = v;
    label_out[e] = lab;
  }
}

}  // namespace icl
