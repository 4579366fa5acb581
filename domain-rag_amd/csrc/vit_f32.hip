// vit_f32.hip — the float32 pieces of the CLIP ViT-B/32 image tower that are not a matrix product (those run on
// conv2d_f32_kernel, csrc/lama.hip: the patch embedding as a stride-P convolution, every Linear as a 1x1 one).
//
// Replaces clip.model.VisionTransformer.forward (openai/CLIP@dcba3cb, un-vendored) as model.encode_image reaches it from
// retrieval/clip100_resnet_style_all_shots.py:171,284,337,948.  openai-CLIP keeps its weights in fp16 on CUDA and in fp32 on the
// CPU (clip.load: `if str(device) == "cpu": model.float()`); float32 is therefore the reference's own arithmetic on its
// CPU-runnable configuration and at least its precision everywhere else.
//
//   vit_prepare_u8 / _f32   uint8 HWC (ToTensor + Normalize as two IEEE divisions) or normalised float NCHW -> NHWC f32 with a
//                           zero 4th channel (the conv kernel reads 16-byte channel groups)
//   clip_embed_ln_kernel    x[b,0] = class_embedding, x[b,1+p] = patch embedding; + positional_embedding; ln_pre
//   layernorm_f32_kernel    one wave per row, two-pass mean / variance in fp32, affine; strided rows (ln_post reads x[:,0])
//   attn_small_f32_kernel   softmax(q k^T / sqrt(d)) v for short sequences (T <= 64 keys, head_dim <= 64): one workgroup per
//                           (image, head), K / V staged once through LDS, one wave per query: lane j scores key j from its register-resident key row,
//                           wave-wide softmax, lane d accumulates output column d from its register-resident value column
#include "drag_common.h"

namespace {

__global__ void vit_prepare_u8_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, long long npix, float m0, float m1, float m2,
                                      float s0, float s1, float s2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const uint8_t* p = img + i * 3;
  const f32x4_t o = {((float)p[0] / 255.0f - m0) / s0, ((float)p[1] / 255.0f - m1) / s1, ((float)p[2] / 255.0f - m2) / s2, 0.f};
  *(f32x4_t*)(out + i * 4) = o;
}

__global__ void vit_prepare_f32_kernel(const float* __restrict__ img, float* __restrict__ out, int B, long long hw) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * hw) return;
  const long long b = i / hw, r = i - b * hw;
  const float* p = img + b * 3 * hw + r;
  const f32x4_t o = {p[0], p[hw], p[2 * hw], 0.f};
  *(f32x4_t*)(out + i * 4) = o;
}

// row statistics of `n` values held v[0..PER) per lane (element e = lane + 64*j); returns (mean, rstd)
template <int PER>
__device__ __forceinline__ void row_norm(float (&v)[PER], int n, int lane, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) s += (lane + 64 * j < n) ? v[j] : 0.f;
  mean = wave_sum(s) / (float)n;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const float d = (lane + 64 * j < n) ? v[j] - mean : 0.f;
    q += d * d;
  }
  rstd = 1.0f / sqrtf(wave_sum(q) / (float)n + eps);
}

constexpr int LN_PER = 16;   // D <= 1024

__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ g,
                                                            const float* __restrict__ b, long long rows, int D, long long ldx, long long ldy,
                                                            float eps) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[LN_PER];
#pragma unroll
  for (int j = 0; j < LN_PER; ++j) v[j] = (lane + 64 * j < D) ? x[row * ldx + lane + 64 * j] : 0.f;
  float mean, rstd;
  row_norm<LN_PER>(v, D, lane, eps, mean, rstd);
#pragma unroll
  for (int j = 0; j < LN_PER; ++j) {
    const int e = lane + 64 * j;
    if (e < D) y[row * ldy + e] = (v[j] - mean) * rstd * g[e] + b[e];
  }
}

__global__ __launch_bounds__(256) void clip_embed_ln_kernel(const float* __restrict__ emb, const float* __restrict__ cls,
                                                            const float* __restrict__ pos, const float* __restrict__ g,
                                                            const float* __restrict__ b, float* __restrict__ x, int B, int T, int D, float eps) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * T) return;
  const int t = (int)(row % T);
  const long long bi = row / T;
  const float* src = t == 0 ? cls : emb + (bi * (T - 1) + (t - 1)) * D;
  float v[LN_PER];
#pragma unroll
  for (int j = 0; j < LN_PER; ++j) {
    const int e = lane + 64 * j;
    v[j] = e < D ? src[e] + pos[(long long)t * D + e] : 0.f;
  }
  float mean, rstd;
  row_norm<LN_PER>(v, D, lane, eps, mean, rstd);
#pragma unroll
  for (int j = 0; j < LN_PER; ++j) {
    const int e = lane + 64 * j;
    if (e < D) x[row * D + e] = (v[j] - mean) * rstd * g[e] + b[e];
  }
}

// the value lane `src` holds, as a wave-uniform (scalar) operand
__device__ __forceinline__ float bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// qkv rows [B*T, ld] with q at column h*hd, k at D + h*hd, v at 2D + h*hd; out rows [B*T, ldo] column h*hd.
// Everything a wave needs for its queries lives in registers: lane j keeps key row j (64 floats) and lane d keeps value
// column d (64 floats, zero beyond T), loaded once per (image, head).  Per query the q element / the probability of key j is
// made wave-uniform with v_readlane and fed to v_fmac as a scalar operand — no LDS traffic and no shuffles in the loops.
__global__ __launch_bounds__(256) void attn_small_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int H, int hd, int ld,
                                                             int ldo, float scale) {
  const int h = blockIdx.x, bi = blockIdx.y;
  const int D = H * hd;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = qkv + (long long)bi * T * ld + h * hd;
  // K and V of this (image, head) pass through LDS once per workgroup (coalesced 16-byte loads; one copy instead of one per
  // wave), then every wave fills its registers from there: lane j <- key row j, lane d <- value column d
  __shared__ float Ks[64][65];
  __shared__ float Vs[64][64];
  const int hq = hd >> 2;
  for (int i = threadIdx.x; i < T * hq; i += 256) {
    const int t = i / hq, d4 = (i - t * hq) * 4;
    const f32x4_t k4 = *(const f32x4_t*)(base + (long long)t * ld + D + d4);
    const f32x4_t v4 = *(const f32x4_t*)(base + (long long)t * ld + 2 * D + d4);
    Ks[t][d4] = k4[0]; Ks[t][d4 + 1] = k4[1]; Ks[t][d4 + 2] = k4[2]; Ks[t][d4 + 3] = k4[3];
    *(f32x4_t*)&Vs[t][d4] = v4;
  }
  __syncthreads();
  float krow[64], vcol[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) krow[d] = (lane < T && d < hd) ? Ks[lane][d] : 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) vcol[j] = (j < T && lane < hd) ? Vs[j][lane] : 0.f;
  for (int t = wave; t < T; t += 4) {
    const float q = lane < hd ? base[(long long)t * ld + lane] : 0.f;
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < 64; ++d) a = fmaf(bcast(q, d), krow[d], a);
    const float s = lane < T ? a * scale : -INFINITY;
    const float m = wave_max(s);
    const float e = lane < T ? expf(s - m) : 0.f;
    const float p = e / wave_sum(e);
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) o = fmaf(bcast(p, j), vcol[j], o);
    if (lane < hd) out[((long long)bi * T + t) * ldo + h * hd + lane] = o;
  }
}

}  // namespace

extern "C" int drag_vit_prepare_u8(const void* img, float* out, int64_t npix, const float* mean3, const float* std3, void* stream) {
  DRAG_CHECK(img && out && mean3 && std3 && npix > 0, "vit_prepare_u8: bad args");
  hipLaunchKernelGGL(vit_prepare_u8_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img, out,
                     (long long)npix, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_vit_prepare_f32(const float* img, float* out, int32_t B, int64_t hw, void* stream) {
  DRAG_CHECK(img && out && B > 0 && hw > 0, "vit_prepare_f32: bad args");
  const long long n = (long long)B * hw;
  hipLaunchKernelGGL(vit_prepare_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, (long long)hw);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_layernorm_f32(const float* x, float* y, const float* gamma, const float* beta, int64_t rows, int32_t D, int64_t ldx,
                                  int64_t ldy, float eps, void* stream) {
  DRAG_CHECK(x && y && gamma && beta, "layernorm_f32: null pointer");
  DRAG_CHECK(rows > 0 && D > 0 && D <= 64 * LN_PER && ldx >= D && ldy >= D, "layernorm_f32: bad shape (D <= 1024)");
  hipLaunchKernelGGL(layernorm_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, gamma, beta,
                     (long long)rows, D, (long long)ldx, (long long)ldy, eps);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_clip_embed_ln_f32(const float* emb, const float* cls, const float* pos, const float* gamma, const float* beta, float* x,
                                      int32_t B, int32_t T, int32_t D, float eps, void* stream) {
  DRAG_CHECK(emb && cls && pos && gamma && beta && x, "clip_embed_ln_f32: null pointer");
  DRAG_CHECK(B > 0 && T > 1 && D > 0 && D <= 64 * LN_PER, "clip_embed_ln_f32: bad shape (D <= 1024)");
  const long long rows = (long long)B * T;
  hipLaunchKernelGGL(clip_embed_ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, emb, cls, pos, gamma, beta, x,
                     B, T, D, eps);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_attention_small_f32(const float* qkv, float* out, int32_t B, int32_t T, int32_t H, int32_t head_dim, int32_t ld,
                                        int32_t ldo, float scale, void* stream) {
  DRAG_CHECK(qkv && out, "attention_small_f32: null pointer");
  DRAG_CHECK(B > 0 && B <= 65535 && H > 0 && T > 0 && T <= 64 && head_dim > 0 && head_dim <= 64, "attention_small_f32: needs T <= 64, head_dim <= 64");
  DRAG_CHECK(ld >= 3 * H * head_dim && ldo >= H * head_dim, "attention_small_f32: row strides");
  DRAG_CHECK(head_dim % 4 == 0 && ld % 4 == 0 && ((uintptr_t)qkv & 15) == 0, "attention_small_f32: head_dim and row stride must be multiples of 4 floats");
  hipLaunchKernelGGL(attn_small_f32_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream, qkv, out, T, H, head_dim, ld, ldo, scale);
  DRAG_LAUNCH_CHECK();
  return 0;
}
