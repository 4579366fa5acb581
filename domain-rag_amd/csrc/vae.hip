// vae.hip — HBM-bound kernels around the Flux VAE convolutions (gfx950): GroupNorm(32)+SiLU over
// NHWC, nearest-2x upsample into a zero-haloed buffer, row softmax for the mid-block attention,
// latent pack/unpack, image pre/post-processing and the Fill mask packing.
//
// Replaces (diffusers 0.33.1, un-vendored): AutoencoderKL's GroupNorm/SiLU/Upsample2D,
// DiagonalGaussianDistribution.sample, FluxPipeline._pack_latents/_unpack_latents,
// FluxFillPipeline.prepare_mask_latents, VaeImageProcessor.preprocess/postprocess — reached from
// pipe(...) / pipe_fill(...) at batch_generate_flux_kshot.py:467-474 and
// outpainting_updown_sampling_redux.py:1246-1257.
//
// Layout: activations are NHWC bf16 (channels innermost = the implicit-GEMM K axis).  Tensors that
// feed a 3x3 convolution carry a 1-pixel zero halo ([B, H+2, W+2, C]); kernels only ever write the
// interior, so the halo stays zero for the lifetime of the buffer.
#include "drag_common.h"

namespace {

// ------------------------------------------------------------------ GroupNorm
// pass 1: per (image, pixel chunk) partial sums for every unit of 4 channels.  Sums are taken of x − c and (x − c)² with a shift c per
// group: E[x²] − E[x]² in float32 partials loses the variance of a group whose mean is large against its spread (mean 100, spread 0.3:
// 1e4 · 1e-6 of rounding against a variance of 0.1), torch's GroupNorm (Welford) does not; any c near the mean removes the cancellation
// and costs two subtractions per element of an HBM-bound pass.  c = the mean of the group's channels at four pixels spread over the image
// (gn_shift: round 4 took the single element (pixel 0, first channel) — one outlier there brought the cancellation back, one Inf there
// turned every statistic of the group into NaN where plain sums would have stayed finite for the other groups).  A
// non-finite shift counts as 0.  Both passes call the same function on the same data: the same float.
struct GnPartArgs {
  const bf16_t* x;   // [B, HW, C]
  float* part;       // [B, nchunks, C/4, 2]
  int HW, C, G, chunk;  // pixels per block
};

__device__ __forceinline__ float gn_shift(const bf16_t* xb, int HW, int C, int cpg, int g) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long px = (long long)k * (HW - 1) / 3;
    for (int c = 0; c < cpg; ++c) s += bf2f(xb[px * C + g * cpg + c]);
  }
  s *= 1.0f / (4.0f * (float)cpg);
  return (s - s == 0.f) ? s : 0.f;          // Inf / NaN -> 0
}

__global__ __launch_bounds__(256) void gn_partial_kernel(GnPartArgs p) {
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x;
  const int slots = p.C / 8;                 // 16-byte slots per pixel (16, 32 or 64)
  const int slot = tid % slots, prow = tid / slots, pstep = 256 / slots;
  const int b = blockIdx.y, ch = blockIdx.x;
  const int p0 = ch * p.chunk, p1 = min(p0 + p.chunk, p.HW);
  const bf16_t* xb = p.x + (long long)b * p.HW * p.C;
  const int cpg = p.C / p.G;                 // channels per group (4, 8 or 16): a unit of 4 channels lies inside one group
  const float sh0 = gn_shift(xb, p.HW, p.C, cpg, (slot * 8) / cpg), sh1 = gn_shift(xb, p.HW, p.C, cpg, (slot * 8 + 4) / cpg);
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  for (int px = p0 + prow; px < p1; px += pstep) {
    const u32x4_t raw = *(const u32x4_t*)(xb + (long long)px * p.C + slot * 8);
    const float a0 = bf2f((bf16_t)(raw[0] & 0xffff)) - sh0, a1 = bf2f((bf16_t)(raw[0] >> 16)) - sh0;
    const float a2 = bf2f((bf16_t)(raw[1] & 0xffff)) - sh0, a3 = bf2f((bf16_t)(raw[1] >> 16)) - sh0;
    const float c0 = bf2f((bf16_t)(raw[2] & 0xffff)) - sh1, c1 = bf2f((bf16_t)(raw[2] >> 16)) - sh1;
    const float c2 = bf2f((bf16_t)(raw[3] & 0xffff)) - sh1, c3 = bf2f((bf16_t)(raw[3] >> 16)) - sh1;
    s0 += (a0 + a1) + (a2 + a3);
    q0 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    s1 += (c0 + c1) + (c2 + c3);
    q1 += (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3);
  }
  red[tid * 4 + 0] = s0; red[tid * 4 + 1] = q0; red[tid * 4 + 2] = s1; red[tid * 4 + 3] = q1;
  __syncthreads();
  // threads with the same slot are tid = slot + k*slots; fixed-order sum -> deterministic
  if (tid < slots) {
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < pstep; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] += red[(tid + k * slots) * 4 + j];
    float* o = p.part + (((long long)b * gridDim.x + ch) * (p.C / 4) + tid * 2) * 2;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
  }
}

// pass 2: per (image, group) mean / rstd, accumulated in double in a fixed order
__global__ void gn_finalize_kernel(const bf16_t* x, const float* part, float* stats, int nchunks, int C, int G, int HW, float eps) {
  const int b = blockIdx.x, g = threadIdx.x;
  if (g >= G) return;
  const int upg = (C / G) / 4;  // units of 4 channels per group
  double s = 0.0, q = 0.0;
  for (int ch = 0; ch < nchunks; ++ch)
    for (int u = 0; u < upg; ++u) {
      const float* o = part + (((long long)b * nchunks + ch) * (C / 4) + g * upg + u) * 2;
      s += o[0]; q += o[1];
    }
  const double n = (double)HW * (C / G);
  const double dm = s / n;                   // mean of x − c
  double var = q / n - dm * dm;
  if (var < 0) var = 0;
  const double mean = (double)gn_shift(x + (long long)b * HW * C, HW, C, C / G, g) + dm;
  stats[(b * G + g) * 2] = (float)mean;
  stats[(b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// pass 3: normalise + affine (+ SiLU), NHWC -> NHWC (optionally into a haloed buffer)
struct GnApplyArgs {
  const bf16_t* x;
  bf16_t* y;
  const bf16_t *gamma, *beta;
  const float* stats;
  int B, H, W, C, G, pad, silu;
};

__global__ __launch_bounds__(256) void gn_apply_kernel(GnApplyArgs p) {
  const int slots = p.C / 8;
  const long long total = (long long)p.B * p.H * p.W * slots;
  const int cpg = p.C / p.G;
  const int Wp = p.W + 2 * p.pad, Hp = p.H + 2 * p.pad;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int slot = (int)(i % slots);
    const long long pix = i / slots;
    const int xw = (int)(pix % p.W);
    const long long t = pix / p.W;
    const int yh = (int)(t % p.H), b = (int)(t / p.H);
    const int c0 = slot * 8;
    const u32x4_t raw = *(const u32x4_t*)(p.x + pix * p.C + c0);
    const u32x4_t gm = *(const u32x4_t*)(p.gamma + c0);
    const u32x4_t bt = *(const u32x4_t*)(p.beta + c0);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ca = c0 + 2 * j, cb = ca + 1;
      const float* sa = p.stats + (b * p.G + ca / cpg) * 2;
      const float* sb = p.stats + (b * p.G + cb / cpg) * 2;
      float va = (bf2f((bf16_t)(raw[j] & 0xffff)) - sa[0]) * sa[1] * bf2f((bf16_t)(gm[j] & 0xffff)) + bf2f((bf16_t)(bt[j] & 0xffff));
      float vb = (bf2f((bf16_t)(raw[j] >> 16)) - sb[0]) * sb[1] * bf2f((bf16_t)(gm[j] >> 16)) + bf2f((bf16_t)(bt[j] >> 16));
      if (p.silu) {  // torch: GroupNorm output is a bf16 tensor before SiLU reads it
        va = act_silu(rbf(va));
        vb = act_silu(rbf(vb));
      }
      o[j] = pack2bf(va, vb);
    }
    const long long opix = ((long long)b * Hp + yh + p.pad) * Wp + xw + p.pad;
    *(u32x4_t*)(p.y + opix * p.C + c0) = o;
  }
}

// ------------------------------------------------------------------ copies into haloed buffers
struct PadCopyArgs {
  const bf16_t* x;  // [B, H, W, C]
  bf16_t* y;        // [B, up*H + 2, up*W + 2, C]
  int B, H, W, C, up;
};
__global__ __launch_bounds__(256) void pad_copy_kernel(PadCopyArgs p) {
  const int slots = p.C / 8;
  const int Ho = p.H * p.up, Wo = p.W * p.up;
  const long long total = (long long)p.B * Ho * Wo * slots;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int slot = (int)(i % slots);
    const long long pix = i / slots;
    const int xo = (int)(pix % Wo);
    const long long t = pix / Wo;
    const int yo = (int)(t % Ho), b = (int)(t / Ho);
    const u32x4_t v = *(const u32x4_t*)(p.x + (((long long)b * p.H + yo / p.up) * p.W + xo / p.up) * p.C + slot * 8);
    *(u32x4_t*)(p.y + (((long long)b * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1) * p.C + slot * 8) = v;
  }
}

// ------------------------------------------------------------------ row softmax (f32 -> bf16)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, bf16_t* y, int cols, float scale, int ldy) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* xr = x + row * cols;
  bf16_t* yr = y + row * ldy;
  for (int c = cols + threadIdx.x; c < ldy; c += 256) yr[c] = 0;   // zero the K-padding the next GEMM reads
  const int tid = threadIdx.x;
  float m = -INFINITY;
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4_t v = *(const f32x4_t*)(xr + c);
    m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4_t v = *(const f32x4_t*)(xr + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) s += __expf((v[j] - m) * scale);
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
  __syncthreads();
  s = (red[4] + red[5]) + (red[6] + red[7]);
  const float inv = 1.0f / s;
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4_t v = *(const f32x4_t*)(xr + c);
    u32x2_t o;
    o[0] = pack2bf(__expf((v[0] - m) * scale) * inv, __expf((v[1] - m) * scale) * inv);
    o[1] = pack2bf(__expf((v[2] - m) * scale) * inv, __expf((v[3] - m) * scale) * inv);
    *(u32x2_t*)(yr + c) = o;
  }
}

// ------------------------------------------------------------------ latent pack / unpack
// tokens [B, h*w, ld] (64 features at column col0) -> haloed NHWC latents [B, 2h+2, 2w+2, C] (16 real ch)
// value = tok / scaling + shift  (bf16 ops like `latents / scaling_factor + shift_factor`)
struct UnpackArgs { const bf16_t* tok; bf16_t* y; int B, h, w, ld, C; float scaling, shift; };
__global__ __launch_bounds__(256) void unpack_latents_kernel(UnpackArgs p) {
  const long long total = (long long)p.B * p.h * p.w * 64;
  const int Hp = 2 * p.h + 2, Wp = 2 * p.w + 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i & 63);
    const long long t = i >> 6;
    const int j = (int)(t % p.w);
    const long long t2 = t / p.w;
    const int ii = (int)(t2 % p.h), b = (int)(t2 / p.h);
    const int c = f >> 2, di = (f >> 1) & 1, dj = f & 1;
    // torch divides a tensor by a scalar as a * (1 / scalar) in fp32 opmath
    const float v = rbf(rbf(bf2f(p.tok[t * p.ld + f]) * (1.0f / p.scaling)) + p.shift);
    p.y[(((long long)b * Hp + 2 * ii + di + 1) * Wp + 2 * j + dj + 1) * p.C + c] = f2bf(v);
  }
}

// encoder moments [B, H, W, ldm] (mean 0..15 | logvar 16..31) + noise NCHW [B,16,H,W] ->
// packed tokens [B, (H/2)(W/2), ld] at column col0:  ((mean + exp(0.5*clamp(logvar)) * noise) - shift) * scaling
struct SamplePackArgs { const bf16_t* mom; const bf16_t* noise; bf16_t* tok; int B, H, W, ldm, ld; float scaling, shift; };
__global__ __launch_bounds__(256) void sample_pack_kernel(SamplePackArgs p) {
  const int h = p.H / 2, w = p.W / 2;
  const long long total = (long long)p.B * h * w * 64;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i & 63);
    const long long t = i >> 6;
    const int j = (int)(t % w);
    const long long t2 = t / w;
    const int ii = (int)(t2 % h), b = (int)(t2 / h);
    const int c = f >> 2, di = (f >> 1) & 1, dj = f & 1;
    const int y = 2 * ii + di, x = 2 * j + dj;
    const bf16_t* m = p.mom + (((long long)b * p.H + y) * p.W + x) * p.ldm;
    const float mean = bf2f(m[c]);
    const float logvar = fminf(fmaxf(bf2f(m[16 + c]), -30.0f), 20.0f);
    const float stdv = rbf(__expf(rbf(0.5f * logvar)));
    float v = mean;
    if (p.noise) v = rbf(mean + rbf(stdv * bf2f(p.noise[(((long long)b * 16 + c) * p.H + y) * p.W + x])));
    v = rbf(rbf(v - p.shift) * p.scaling);
    p.tok[t * p.ld + f] = f2bf(v);
  }
}

// ------------------------------------------------------------------ image pre / post
// uint8 RGB [B, H, W, 3] (+ optional uint8 mask [B, H, W], 255/>=128 = repaint) -> haloed NHWC [B,H+2,W+2,C]
// value = (u8/255)*2-1, times (1 - mask) when a mask is given (FluxFillPipeline: masked_image)
struct PreArgs { const uint8_t* img; const uint8_t* mask; bf16_t* y; int B, H, W, C; };
__global__ __launch_bounds__(256) void image_preprocess_kernel(PreArgs p) {
  const long long total = (long long)p.B * p.H * p.W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % p.W);
    const long long t = i / p.W;
    const int y = (int)(t % p.H), b = (int)(t / p.H);
    float keep = 1.0f;
    if (p.mask) keep = (p.mask[i] / 255.0f) < 0.5f ? 1.0f : 0.0f;
    bf16_t* o = p.y + (((long long)b * (p.H + 2) + y + 1) * (p.W + 2) + x + 1) * p.C;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = (2.0f * (p.img[i * 3 + c] / 255.0f) - 1.0f) * keep;
      o[c] = f2bf(v);
    }
  }
}

// decoder output rows [B*H*W, ld] bf16 (3 real channels) -> uint8 RGB: ((x/2+0.5).clamp(0,1)*255).round()
__global__ __launch_bounds__(256) void image_postprocess_kernel(const bf16_t* x, uint8_t* out, long long npix, int ld) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = rbf(bf2f(x[i * ld + c]) * 0.5f + 0.5f);
      v = fminf(fmaxf(v, 0.0f), 1.0f);
      out[i * 3 + c] = (uint8_t)rintf(v * 255.0f);
    }
  }
}

// Fill mask: uint8 [B, H, W] -> tokens [B, (H/16)(W/16), ld] 256 features at col0:
// f = (dy*8+dx)*4 + di*2 + dj  <-  binarise(mask[b, (2i+di)*8+dy, (2j+dj)*8+dx])
struct MaskPackArgs { const uint8_t* mask; bf16_t* tok; int B, H, W, ld; };
__global__ __launch_bounds__(256) void mask_pack_kernel(MaskPackArgs p) {
  const int h = p.H / 16, w = p.W / 16;
  const long long total = (long long)p.B * h * w * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i & 255);
    const long long t = i >> 8;
    const int j = (int)(t % w);
    const long long t2 = t / w;
    const int ii = (int)(t2 % h), b = (int)(t2 / h);
    const int chn = f >> 2, di = (f >> 1) & 1, dj = f & 1;
    const int dy = chn >> 3, dx = chn & 7;
    const int y = (2 * ii + di) * 8 + dy, x = (2 * j + dj) * 8 + dx;
    const float m = (p.mask[((long long)b * p.H + y) * p.W + x] / 255.0f) < 0.5f ? 0.0f : 1.0f;
    p.tok[t * p.ld + f] = f2bf(m);
  }
}

// strided flow-Euler step and noise mixing on token rows
struct RowsArgs { bf16_t* x; const bf16_t* v; long long rows; int cols, ldx, ldv; float a; };
__global__ __launch_bounds__(256) void euler_rows_kernel(RowsArgs p) {
  const long long total = p.rows * p.cols;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / p.cols;
    const int c = (int)(i - r * p.cols);
    bf16_t* xp = p.x + r * p.ldx + c;
    *xp = f2bf(bf2f(*xp) + p.a * bf2f(p.v[r * p.ldv + c]));
  }
}
// FlowMatchEulerDiscreteScheduler.scale_noise in bf16: x = sigma*noise + (1-sigma)*x
__global__ __launch_bounds__(256) void scale_noise_rows_kernel(RowsArgs p) {
  const long long total = p.rows * p.cols;
  const float sg = rbf(p.a), om = rbf(1.0f - rbf(p.a));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / p.cols;
    const int c = (int)(i - r * p.cols);
    bf16_t* xp = p.x + r * p.ldx + c;
    *xp = f2bf(rbf(sg * bf2f(p.v[r * p.ldv + c])) + rbf(om * bf2f(*xp)));
  }
}

inline int grid_for(long long n) {
  long long g = (n + 255) / 256;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (int)g;
}

}  // namespace

extern "C" int64_t drag_groupnorm_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C) {
  const long long HW = (long long)H * W;
  const long long nchunks = (HW + 1023) / 1024;
  return B * nchunks * (C / 4) * 2 * 4 + (long long)B * 32 * 2 * 4 + 256;
}

extern "C" int drag_groupnorm_silu_bf16(const void* x, void* y, const void* gamma, const void* beta, int32_t B, int32_t H,
                                        int32_t W, int32_t C, int32_t groups, int32_t out_pad, int32_t silu, float eps,
                                        void* workspace, void* stream) {
  DRAG_CHECK(x && y && gamma && beta && workspace, "drag_groupnorm_silu_bf16: null pointer");
  DRAG_CHECK(B > 0 && H > 0 && W > 0, "drag_groupnorm_silu_bf16: bad shape");
  DRAG_CHECK(groups == 32 && (C == 128 || C == 256 || C == 512), "drag_groupnorm_silu_bf16: 32 groups, C in {128,256,512}");
  DRAG_CHECK(out_pad == 0 || out_pad == 1, "drag_groupnorm_silu_bf16: out_pad 0 or 1");
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  const int nchunks = (HW + 1023) / 1024;
  float* part = (float*)workspace;
  float* stats = part + (long long)B * nchunks * (C / 4) * 2;
  GnPartArgs pa{(const bf16_t*)x, part, HW, C, groups, 1024};
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunks, B), dim3(256), 0, st, pa);
  DRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(64), 0, st, (const bf16_t*)x, (const float*)part, stats, nchunks, C, groups, HW, eps);
  DRAG_LAUNCH_CHECK();
  GnApplyArgs aa{(const bf16_t*)x, (bf16_t*)y, (const bf16_t*)gamma, (const bf16_t*)beta, stats, B, H, W, C, groups, out_pad, silu};
  hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for((long long)B * HW * (C / 8))), dim3(256), 0, st, aa);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_pad_copy_bf16(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t upsample,
                                  void* stream) {
  DRAG_CHECK(x && y && B > 0 && H > 0 && W > 0 && C % 8 == 0, "drag_pad_copy_bf16: bad args");
  DRAG_CHECK(upsample == 1 || upsample == 2, "drag_pad_copy_bf16: upsample 1 or 2");
  PadCopyArgs p{(const bf16_t*)x, (bf16_t*)y, B, H, W, C, upsample};
  hipLaunchKernelGGL(pad_copy_kernel, dim3(grid_for((long long)B * H * W * upsample * upsample * (C / 8))), dim3(256), 0,
                     (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_softmax_rows_f32_bf16(const float* x, void* y, int64_t rows, int32_t cols, int32_t ldy, float scale,
                                          void* stream) {
  DRAG_CHECK(x && y && rows > 0 && cols > 0 && cols % 4 == 0 && ldy >= cols && ldy % 4 == 0,
             "drag_softmax_rows_f32_bf16: bad args");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, cols, scale, ldy);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_unpack_latents_bf16(const void* tokens, void* y, int32_t B, int32_t h, int32_t w, int32_t ld, int32_t C,
                                        float scaling, float shift, void* stream) {
  DRAG_CHECK(tokens && y && B > 0 && h > 0 && w > 0 && C >= 16, "drag_unpack_latents_bf16: bad args");
  UnpackArgs p{(const bf16_t*)tokens, (bf16_t*)y, B, h, w, ld, C, scaling, shift};
  hipLaunchKernelGGL(unpack_latents_kernel, dim3(grid_for((long long)B * h * w * 64)), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_sample_pack_latents_bf16(const void* moments, const void* noise, void* tokens, int32_t B, int32_t H,
                                             int32_t W, int32_t ldm, int32_t ld, float scaling, float shift, void* stream) {
  DRAG_CHECK(moments && tokens && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && ldm >= 32,
             "drag_sample_pack_latents_bf16: bad args");
  SamplePackArgs p{(const bf16_t*)moments, (const bf16_t*)noise, (bf16_t*)tokens, B, H, W, ldm, ld, scaling, shift};
  hipLaunchKernelGGL(sample_pack_kernel, dim3(grid_for((long long)B * H * W * 16)), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_image_preprocess_u8(const void* img, const void* mask, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                                        void* stream) {
  DRAG_CHECK(img && y && B > 0 && H > 0 && W > 0 && C >= 3, "drag_image_preprocess_u8: bad args");
  PreArgs p{(const uint8_t*)img, (const uint8_t*)mask, (bf16_t*)y, B, H, W, C};
  hipLaunchKernelGGL(image_preprocess_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_image_postprocess_u8(const void* x, void* out, int64_t npix, int32_t ld, void* stream) {
  DRAG_CHECK(x && out && npix > 0 && ld >= 3, "drag_image_postprocess_u8: bad args");
  hipLaunchKernelGGL(image_postprocess_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (uint8_t*)out, (long long)npix, ld);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_mask_pack_u8(const void* mask, void* tokens, int32_t B, int32_t H, int32_t W, int32_t ld, void* stream) {
  DRAG_CHECK(mask && tokens && B > 0 && H % 16 == 0 && W % 16 == 0 && H > 0 && W > 0, "drag_mask_pack_u8: bad args");
  MaskPackArgs p{(const uint8_t*)mask, (bf16_t*)tokens, B, H, W, ld};
  hipLaunchKernelGGL(mask_pack_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_flow_euler_rows_bf16(void* x, const void* v, int64_t rows, int32_t cols, int32_t ldx, int32_t ldv,
                                         float dt, void* stream) {
  DRAG_CHECK(x && v && rows > 0 && cols > 0, "drag_flow_euler_rows_bf16: bad args");
  RowsArgs p{(bf16_t*)x, (const bf16_t*)v, rows, cols, ldx, ldv, dt};
  hipLaunchKernelGGL(euler_rows_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_scale_noise_rows_bf16(void* x, const void* noise, int64_t rows, int32_t cols, int32_t ldx, int32_t ldn,
                                          float sigma, void* stream) {
  DRAG_CHECK(x && noise && rows > 0 && cols > 0, "drag_scale_noise_rows_bf16: bad args");
  RowsArgs p{(bf16_t*)x, (const bf16_t*)noise, rows, cols, ldx, ldn, sigma};
  hipLaunchKernelGGL(scale_noise_rows_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}
