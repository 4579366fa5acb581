// stem.hip — ResNet50-stem "style" statistics for the second-stage re-rank (gfx950, fp32).
//
// Replaces ResNetEncoder.forward (conv1 7x7/2 + bn1(eval) + relu + maxpool 3x3/2) + calc_mean_std
// (retrieval/clip100_resnet_style_all_shots.py:51-74,197-200): input [B,3,H,W] fp32 in [0,1] with NO
// ImageNet normalisation, output [B, 128] = concat(channel mean, sqrt(unbiased var + 1e-5)).
// Two kernels: (A) one block per 8x8 tile of the pooled map x all 64 channels, input tile + transposed weights in
// LDS (inner loop = broadcast LDS read + conflict-free LDS read + fmaf, k-order ch,ky,kx like a direct conv);
// (B) one block per (image, channel): two-pass mean / unbiased variance of the pooled map.
#include "drag_common.h"

namespace {

struct StemArgs {
  const float* img;     // [B, 3, H, W]
  const float* w;       // [64, 3, 7, 7]
  const float *bn_scale, *bn_shift;  // folded eval BatchNorm: y = conv * scale + shift
  float* pooled;        // scratch [B, 64, Hp, Wp]
  float* out;           // [B, 128]
  int B, H, W, Hc, Wc, Hp, Wp;
  float eps;
};

constexpr int ST_T = 8;                        // pooled outputs per tile edge
constexpr int ST_IN = 4 * (ST_T - 1) + 4 + 7;  // 39 input rows/cols feed an 8x8 pooled tile
constexpr int ST_INP = ST_IN + 1;              // row pitch

// kernel A: conv1 + bn + relu + maxpool for one 8x8 pooled tile x 64 channels.  The input tile (with its zero
// halo) and the transposed weights [tap][channel] live in LDS: the inner loop is one broadcast LDS read (input,
// same address for the 64 channel-lanes of a wave), one conflict-free LDS read (weight) and one fmaf.
__global__ __launch_bounds__(256) void stem_pool_kernel(StemArgs p) {
  __shared__ float s_in[3 * ST_IN * ST_INP];
  __shared__ float s_w[147 * 64];
  const int tid = threadIdx.x;
  const int tiles_x = (p.Wp + ST_T - 1) / ST_T;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x, b = blockIdx.y;
  const int py0 = ty * ST_T, px0 = tx * ST_T;
  const int iy0 = 4 * py0 - 5, ix0 = 4 * px0 - 5;
  const float* im = p.img + (long long)b * 3 * p.H * p.W;
  for (int i = tid; i < 3 * ST_IN * ST_IN; i += 256) {
    const int ch = i / (ST_IN * ST_IN), r = i - ch * ST_IN * ST_IN;
    const int y = r / ST_IN, x = r - y * ST_IN;
    const int iy = iy0 + y, ix = ix0 + x;
    float v = 0.f;                                                   // conv zero padding
    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = im[((long long)ch * p.H + iy) * p.W + ix];
    s_in[(ch * ST_IN + y) * ST_INP + x] = v;
  }
  for (int i = tid; i < 147 * 64; i += 256) {
    const int c = i & 63, tap = i >> 6;
    s_w[tap * 64 + c] = p.w[c * 147 + tap];
  }
  __syncthreads();
  const int c = tid & 63, g = tid >> 6;
  const float sc = p.bn_scale[c], sh = p.bn_shift[c];
  for (int i = 0; i < 16; ++i) {
    const int pl = g * 16 + i, pyl = pl >> 3, pxl = pl & 7;
    const int py = py0 + pyl, px = px0 + pxl;
    if (py >= p.Hp || px >= p.Wp) continue;
    float best = -INFINITY;                                          // maxpool pads with -inf
    for (int dy = 0; dy < 3; ++dy) {
      const int cy = 2 * py - 1 + dy;
      if (cy < 0 || cy >= p.Hc) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int cx = 2 * px - 1 + dx;
        if (cx < 0 || cx >= p.Wc) continue;
        float acc = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
          for (int ky = 0; ky < 7; ++ky) {
            const float* row = s_in + (ch * ST_IN + 4 * pyl + 2 * dy + ky) * ST_INP + 4 * pxl + 2 * dx;
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) acc = fmaf(row[kx], s_w[((ch * 7 + ky) * 7 + kx) * 64 + c], acc);
          }
        best = fmaxf(best, fmaxf(acc * sc + sh, 0.f));
      }
    }
    p.pooled[(((long long)b * 64 + c) * p.Hp + py) * p.Wp + px] = best;
  }
}

// kernel B: per (image, channel) mean and unbiased std of the pooled map (two passes, fixed order)
__global__ __launch_bounds__(256) void stem_stats_kernel(StemArgs p) {
  __shared__ float red[8];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int n = p.Hp * p.Wp;
  const float* v = p.pooled + ((long long)b * 64 + c) * n;
  float sum = 0.f;
  for (int i = tid; i < n; i += 256) sum += v[i];
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
  float sq = 0.f;
  for (int i = tid; i < n; i += 256) {
    const float d = v[i] - mean;
    sq += d * d;
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

extern "C" int64_t drag_resnet_stem_style_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;
  const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
  return (int64_t)B * 64 * Hp * Wp * 4;
}

extern "C" int drag_resnet_stem_style_f32(const float* img, const float* conv_w, const float* bn_scale, const float* bn_shift,
                                          float* out, int32_t B, int32_t H, int32_t W, float eps, void* workspace,
                                          void* stream) {
  DRAG_CHECK(img && conv_w && bn_scale && bn_shift && out && workspace, "drag_resnet_stem_style_f32: null pointer");
  DRAG_CHECK(B > 0 && H >= 8 && W >= 8, "drag_resnet_stem_style_f32: bad shape");
  StemArgs p;
  p.img = img; p.w = conv_w; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.pooled = (float*)workspace; p.out = out;
  p.B = B; p.H = H; p.W = W; p.eps = eps;
  p.Hc = (H - 1) / 2 + 1; p.Wc = (W - 1) / 2 + 1;
  p.Hp = (p.Hc - 1) / 2 + 1; p.Wp = (p.Wc - 1) / 2 + 1;
  DRAG_CHECK(p.Hp * p.Wp > 1, "drag_resnet_stem_style_f32: image too small");
  const int tiles = ((p.Hp + ST_T - 1) / ST_T) * ((p.Wp + ST_T - 1) / ST_T);
  hipLaunchKernelGGL(stem_pool_kernel, dim3(tiles, B), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(stem_stats_kernel, dim3(64, B), dim3(256), 0, (hipStream_t)stream, p);
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
