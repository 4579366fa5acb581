// stem.hip — ResNet50-stem "style" statistics for the second-stage re-rank (gfx950, fp32).
//
// Replaces ResNetEncoder.forward (conv1 7x7/2 + bn1(eval) + relu + maxpool 3x3/2) + calc_mean_std
// (retrieval/clip100_resnet_style_all_shots.py:51-74,197-200): input [B,3,H,W] fp32 in [0,1] with NO
// ImageNet normalisation, output [B, 128] = concat(channel mean, sqrt(unbiased var + 1e-5)).
// One block per (image, channel): the 64x64 pooled map of a channel is produced and reduced without
// ever touching HBM (two-pass mean / variance held in registers).  ~0.7 GFLOP per image: latency-,
// not throughput-, bound; the image (786 KB) is read through L2 by the 64 channel blocks.
#include "drag_common.h"

namespace {

struct StemArgs {
  const float* img;     // [B, 3, H, W]
  const float* w;       // [64, 3, 7, 7]
  const float *bn_scale, *bn_shift;  // folded eval BatchNorm: y = conv * scale + shift
  float* out;           // [B, 128]
  int B, H, W;
  float eps;
};

constexpr int MAX_PER_THREAD = 16;  // pooled outputs per thread: (H/4)*(W/4) <= 256*16

__global__ __launch_bounds__(256) void stem_style_kernel(StemArgs p) {
  __shared__ float sw[147];
  __shared__ float red[8];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  if (tid < 147) sw[tid] = p.w[c * 147 + tid];
  __syncthreads();
  const int Hc = (p.H + 2 * 3 - 7) / 2 + 1, Wc = (p.W + 2 * 3 - 7) / 2 + 1;   // conv output
  const int Hp = (Hc + 2 * 1 - 3) / 2 + 1, Wp = (Wc + 2 * 1 - 3) / 2 + 1;     // pooled output
  const int n = Hp * Wp;
  const float sc = p.bn_scale[c], sh = p.bn_shift[c];
  const float* im = p.img + (long long)b * 3 * p.H * p.W;
  float vals[MAX_PER_THREAD];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < MAX_PER_THREAD; ++it) {
    const int o = it * 256 + tid;
    float best = -INFINITY;
    if (o < n) {
      const int py = o / Wp, px = o - py * Wp;
      for (int dy = 0; dy < 3; ++dy) {
        const int cy = 2 * py - 1 + dy;
        if (cy < 0 || cy >= Hc) continue;
        for (int dx = 0; dx < 3; ++dx) {
          const int cx = 2 * px - 1 + dx;
          if (cx < 0 || cx >= Wc) continue;
          float acc = 0.f;
          for (int ch = 0; ch < 3; ++ch)
            for (int ky = 0; ky < 7; ++ky) {
              const int iy = 2 * cy - 3 + ky;
              if (iy < 0 || iy >= p.H) continue;
              for (int kx = 0; kx < 7; ++kx) {
                const int ix = 2 * cx - 3 + kx;
                if (ix < 0 || ix >= p.W) continue;
                acc = fmaf(im[((long long)ch * p.H + iy) * p.W + ix], sw[(ch * 7 + ky) * 7 + kx], acc);
              }
            }
          best = fmaxf(best, fmaxf(acc * sc + sh, 0.f));
        }
      }
      sum += best;
    }
    vals[it] = best;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < MAX_PER_THREAD; ++it) {
    const int o = it * 256 + tid;
    if (o < n) {
      const float d = vals[it] - mean;
      sq += d * d;
    }
  }
  sq = wave_sum(sq);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sq;
  __syncthreads();
  if (tid == 0) {
    const float var = ((red[4] + red[5]) + (red[6] + red[7])) / (float)(n - 1);   // torch .var(): unbiased
    p.out[b * 128 + c] = mean;
    p.out[b * 128 + 64 + c] = sqrtf(var + p.eps);
  }
}

// float NCHW (already normalised) -> patch rows, k = c*P*P + py*P + px (clip preprocess output path)
struct PatchFArgs { const float* img; bf16_t* out; int B, H, W, P, ldo; };
__global__ __launch_bounds__(256) void patchify_f32_kernel(PatchFArgs p) {
  const int gh = p.H / p.P, gw = p.W / p.P;
  const long long total = (long long)p.B * gh * gw * p.ldo;
  const int kk = 3 * p.P * p.P;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % p.ldo);
    const long long row = i / p.ldo;
    float v = 0.f;
    if (k < kk) {
      const int c = k / (p.P * p.P), r = k - c * p.P * p.P;
      const int py = r / p.P, px = r - py * p.P;
      const int pw = (int)(row % gw);
      const long long t = row / gw;
      const int ph = (int)(t % gh), b = (int)(t / gh);
      v = p.img[(((long long)b * 3 + c) * p.H + ph * p.P + py) * p.W + pw * p.P + px];
    }
    p.out[i] = f2bf(v);
  }
}

}  // namespace

extern "C" int drag_resnet_stem_style_f32(const float* img, const float* conv_w, const float* bn_scale, const float* bn_shift,
                                          float* out, int32_t B, int32_t H, int32_t W, float eps, void* stream) {
  DRAG_CHECK(img && conv_w && bn_scale && bn_shift && out, "drag_resnet_stem_style_f32: null pointer");
  DRAG_CHECK(B > 0 && H >= 8 && W >= 8, "drag_resnet_stem_style_f32: bad shape");
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;
  const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
  DRAG_CHECK(Hp * Wp <= 256 * MAX_PER_THREAD && Hp * Wp > 1, "drag_resnet_stem_style_f32: image too large (<= 256x256)");
  StemArgs p{img, conv_w, bn_scale, bn_shift, out, B, H, W, eps};
  hipLaunchKernelGGL(stem_style_kernel, dim3(64, B), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_patchify_f32_nchw(const float* img, void* out, int32_t B, int32_t H, int32_t W, int32_t P, int32_t ldo,
                                      void* stream) {
  DRAG_CHECK(img && out && B > 0 && P > 0 && H >= P && W >= P && ldo >= 3 * P * P, "drag_patchify_f32_nchw: bad args");
  PatchFArgs p{img, (bf16_t*)out, B, H, W, P, ldo};
  long long total = (long long)B * (H / P) * (W / P) * ldo;
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(patchify_f32_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}
