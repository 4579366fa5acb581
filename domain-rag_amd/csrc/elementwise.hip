// elementwise.hip — HBM-bound row/elementwise kernels of the Flux / ViT paths (gfx950).
// All loads/stores are 16 B per lane; one wave owns one row for the normalisations.
//
// Replaces (diffusers 0.33.1 / torch, un-vendored; reached from batch_generate_flux_kshot.py:467-474
// and outpainting_updown_sampling_redux.py:1246-1257): AdaLayerNormZero / AdaLayerNormZeroSingle /
// AdaLayerNormContinuous modulation, nn.LayerNorm, SiLU/GELU, get_timestep_embedding,
// FlowMatchEulerDiscreteScheduler.step.
#include "drag_common.h"

namespace {

constexpr int LN_MAX_IT = 8;  // D <= 64 lanes * 8 elems * 8 = 4096

struct LnArgs {
  const bf16_t* x;
  bf16_t* y;
  const bf16_t *scale, *shift, *gamma, *beta;
  int M, D, ldx, rpb, ldy, ld_mod;
  long long x_bs;
  float eps;
};

// AdaLN fast path for D = NIT * 512 (the DiT's 3072): the generic kernel below guards every 16-byte piece with `c < D`, which
// makes each piece its own basic block — hipcc then waits for each load before issuing the next (six serial round trips per row
// for x, six more for scale / shift).  Here every load of a phase is issued before the first use: x up front, scale / shift while
// the statistics are reduced.  Same arithmetic in the same order as the generic kernel: identical bits.
template <int NIT>
__global__ __launch_bounds__(256) void layernorm_modulate_fixed_kernel(LnArgs p) {
  const int w = wave_id(), l = lane_id();
  const int row = blockIdx.x * 4 + w;
  if (row >= p.M) return;
  const int b = row / p.rpb, s = row - b * p.rpb;
  const bf16_t* xr = p.x + (long long)b * p.x_bs + (long long)s * p.ldx;
  u32x4_t raw[NIT], sc[NIT], sh[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) raw[it] = *(const u32x4_t*)(xr + (it * 64 + l) * 8);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    sc[it] = *(const u32x4_t*)(p.scale + (long long)b * p.ld_mod + (it * 64 + l) * 8);
    sh[it] = *(const u32x4_t*)(p.shift + (long long)b * p.ld_mod + (it * 64 + l) * 8);
  }
  float v[NIT][8];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[it][2 * j] = bf2f((bf16_t)(raw[it][j] & 0xffff));
      v[it][2 * j + 1] = bf2f((bf16_t)(raw[it][j] >> 16));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[it][j];
  }
  const float mean = wave_sum(sum) / (float)p.D;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[it][j] - mean;
      sq += d * d;
    }
  const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);
  bf16_t* yr = p.y + (long long)row * p.ldy;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    u32x4_t pk;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // torch bf16 graph: n = LN(x) (bf16); t = 1 + scale (bf16); y = n * t (bf16) + shift (bf16)
      const float n0 = rbf((v[it][2 * j] - mean) * rstd), n1 = rbf((v[it][2 * j + 1] - mean) * rstd);
      const float t0 = rbf(1.0f + bf2f((bf16_t)(sc[it][j] & 0xffff))), t1 = rbf(1.0f + bf2f((bf16_t)(sc[it][j] >> 16)));
      pk[j] = pack2bf(rbf(n0 * t0) + bf2f((bf16_t)(sh[it][j] & 0xffff)), rbf(n1 * t1) + bf2f((bf16_t)(sh[it][j] >> 16)));
    }
    *(u32x4_t*)(yr + (it * 64 + l) * 8) = pk;
  }
}

__global__ __launch_bounds__(256) void layernorm_modulate_kernel(LnArgs p) {
  const int w = wave_id(), l = lane_id();
  const int row = blockIdx.x * 4 + w;
  if (row >= p.M) return;
  const int b = row / p.rpb, s = row - b * p.rpb;
  const bf16_t* xr = p.x + (long long)b * p.x_bs + (long long)s * p.ldx;
  float v[LN_MAX_IT][8];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    const int c = (it * 64 + l) * 8;
    if (c < p.D) {
      const u32x4_t raw = *(const u32x4_t*)(xr + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[it][2 * j] = bf2f((bf16_t)(raw[j] & 0xffff));
        v[it][2 * j + 1] = bf2f((bf16_t)(raw[j] >> 16));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[it][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)p.D;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    const int c = (it * 64 + l) * 8;
    if (c < p.D) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[it][j] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);
  bf16_t* yr = p.y + (long long)row * p.ldy;
#pragma unroll
  for (int it = 0; it < LN_MAX_IT; ++it) {
    const int c = (it * 64 + l) * 8;
    if (c < p.D) {
      float o[8];
      if (p.gamma) {
        const u32x4_t g = *(const u32x4_t*)(p.gamma + c);
        const u32x4_t be = *(const u32x4_t*)(p.beta + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[2 * j] = (v[it][2 * j] - mean) * rstd * bf2f((bf16_t)(g[j] & 0xffff)) + bf2f((bf16_t)(be[j] & 0xffff));
          o[2 * j + 1] = (v[it][2 * j + 1] - mean) * rstd * bf2f((bf16_t)(g[j] >> 16)) + bf2f((bf16_t)(be[j] >> 16));
        }
      } else if (p.scale) {
        const u32x4_t sc = *(const u32x4_t*)(p.scale + (long long)b * p.ld_mod + c);
        const u32x4_t sh = *(const u32x4_t*)(p.shift + (long long)b * p.ld_mod + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // torch bf16 graph: n = LN(x) (bf16); t = 1 + scale (bf16); y = n * t (bf16) + shift (bf16)
          const float n0 = rbf((v[it][2 * j] - mean) * rstd), n1 = rbf((v[it][2 * j + 1] - mean) * rstd);
          const float t0 = rbf(1.0f + bf2f((bf16_t)(sc[j] & 0xffff))), t1 = rbf(1.0f + bf2f((bf16_t)(sc[j] >> 16)));
          o[2 * j] = rbf(n0 * t0) + bf2f((bf16_t)(sh[j] & 0xffff));
          o[2 * j + 1] = rbf(n1 * t1) + bf2f((bf16_t)(sh[j] >> 16));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[it][j] - mean) * rstd;
      }
      u32x4_t pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) pk[j] = pack2bf(o[2 * j], o[2 * j + 1]);
      *(u32x4_t*)(yr + c) = pk;
    }
  }
}

__global__ __launch_bounds__(256) void act_kernel(const bf16_t* x, bf16_t* y, long long n8, long long n, int act) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    const u32x4_t raw = *(const u32x4_t*)(x + i * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack2bf(apply_act(bf2f((bf16_t)(raw[j] & 0xffff)), act), apply_act(bf2f((bf16_t)(raw[j] >> 16)), act));
    *(u32x4_t*)(y + i * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n8 * 8)) {
    const long long i = n8 * 8 + threadIdx.x;
    y[i] = f2bf(apply_act(bf2f(x[i]), act));
  }
}

__global__ __launch_bounds__(256) void euler_kernel(bf16_t* x, const bf16_t* v, float dt, long long n8, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    const u32x4_t xr = *(const u32x4_t*)(x + i * 8);
    const u32x4_t vr = *(const u32x4_t*)(v + i * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack2bf(bf2f((bf16_t)(xr[j] & 0xffff)) + dt * bf2f((bf16_t)(vr[j] & 0xffff)),
                     bf2f((bf16_t)(xr[j] >> 16)) + dt * bf2f((bf16_t)(vr[j] >> 16)));
    *(u32x4_t*)(x + i * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n8 * 8)) {
    const long long i = n8 * 8 + threadIdx.x;
    x[i] = f2bf(bf2f(x[i]) + dt * bf2f(v[i]));
  }
}

__global__ __launch_bounds__(256) void add_kernel(const bf16_t* a, const bf16_t* b, bf16_t* y, long long n8, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    const u32x4_t ar = *(const u32x4_t*)(a + i * 8);
    const u32x4_t br = *(const u32x4_t*)(b + i * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack2bf(bf2f((bf16_t)(ar[j] & 0xffff)) + bf2f((bf16_t)(br[j] & 0xffff)),
                     bf2f((bf16_t)(ar[j] >> 16)) + bf2f((bf16_t)(br[j] >> 16)));
    *(u32x4_t*)(y + i * 8) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n8 * 8)) {
    const long long i = n8 * 8 + threadIdx.x;
    y[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
  }
}

__global__ __launch_bounds__(256) void cast_f2b_kernel(const float* x, bf16_t* y, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = f2bf(x[i]);
}
__global__ __launch_bounds__(256) void cast_b2f_kernel(const bf16_t* x, float* y, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = bf2f(x[i]);
}

// get_timestep_embedding(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0, scale=1):
// half = dim/2; f_i = exp(-ln(10000) * i / half); emb = [cos(t f) | sin(t f)]  (fp32 math, bf16 out)
__global__ void timestep_embedding_kernel(const float* t, bf16_t* out, int B, int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float f = expf(-9.210340371976184f * (float)j / (float)half);
  const float a = t[b] * f;
  out[(long long)b * dim + j] = f2bf(cosf(a));
  out[(long long)b * dim + half + j] = f2bf(sinf(a));
}

// uint8 RGB [B, H, W, 3] -> normalised patch rows [B*(H/P)*(W/P), ldo] bf16, k = c*P*P + py*P + px
// (the flattening of a Conv2d(3, D, P, stride=P) weight), columns >= 3*P*P zero-filled.
struct PatchArgs { const uint8_t* img; bf16_t* out; int B, H, W, P, ldo; float mean[3], std[3]; };
__global__ __launch_bounds__(256) void patchify_kernel(PatchArgs p) {
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
      const uint8_t u = p.img[(((long long)b * p.H + ph * p.P + py) * p.W + pw * p.P + px) * 3 + c];
      // ToTensor then Normalize as torch computes them: two IEEE divisions (u * (1/255) differs in 126 of 256 values)
      v = ((float)u / 255.0f - p.mean[c]) / p.std[c];
    }
    p.out[i] = f2bf(v);
  }
}

// out[g, e] = bf16( sum_n bf16(scale[g*N+n] * x[g, n, e]) )   (Redux: prompt_embeds *= scale; sum(dim=0))
__global__ __launch_bounds__(256) void scale_sum_kernel(const bf16_t* x, const float* scales, bf16_t* out, int G, int N,
                                                        long long elems) {
  const long long total = (long long)G * elems;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long g = i / elems, e = i - g * elems;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += rbf(rbf(scales[g * N + n]) * bf2f(x[(g * N + n) * elems + e]));
    out[i] = f2bf(acc);
  }
}

inline int ew_grid(long long n8) {
  long long g = (n8 + 255) / 256;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

}  // namespace

extern "C" int drag_layernorm_modulate_bf16(const void* x, void* y, const void* scale, const void* shift,
                                            const void* gamma, const void* beta, int32_t M, int32_t D, int32_t ldx,
                                            int32_t rows_per_batch, int64_t x_batch_stride, int32_t ldy,
                                            int32_t ld_mod, float eps, void* stream) {
  DRAG_CHECK(x && y, "drag_layernorm_modulate_bf16: null pointer");
  DRAG_CHECK(M > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAX_IT, "drag_layernorm_modulate_bf16: D %% 8 == 0, D <= 4096");
  DRAG_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && ld_mod % 8 == 0, "drag_layernorm_modulate_bf16: strides must be multiples of 8");
  DRAG_CHECK((scale == nullptr) == (shift == nullptr) && (gamma == nullptr) == (beta == nullptr),
             "drag_layernorm_modulate_bf16: scale/shift and gamma/beta come in pairs");
  LnArgs p;
  p.x = (const bf16_t*)x; p.y = (bf16_t*)y;
  p.scale = (const bf16_t*)scale; p.shift = (const bf16_t*)shift;
  p.gamma = (const bf16_t*)gamma; p.beta = (const bf16_t*)beta;
  p.M = M; p.D = D; p.ldx = ldx; p.rpb = rows_per_batch > 0 ? rows_per_batch : M; p.ldy = ldy; p.ld_mod = ld_mod;
  p.x_bs = x_batch_stride; p.eps = eps;
  if (D == 6 * 512 && scale && !gamma && !drag_opt(DRAG_OPT_LN_GENERIC))
    hipLaunchKernelGGL(layernorm_modulate_fixed_kernel<6>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(layernorm_modulate_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_act_bf16(const void* x, void* y, int64_t n, int32_t act, void* stream) {
  DRAG_CHECK(x && y && n >= 0, "drag_act_bf16: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y,
                     (long long)(n / 8), (long long)n, act);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_flow_euler_step_bf16(void* x, const void* v, float dt, int64_t n, void* stream) {
  DRAG_CHECK(x && v && n >= 0, "drag_flow_euler_step_bf16: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(euler_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, (const bf16_t*)v,
                     dt, (long long)(n / 8), (long long)n);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
  DRAG_CHECK(a && b && y && n >= 0, "drag_add_bf16: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                     (const bf16_t*)b, (bf16_t*)y, (long long)(n / 8), (long long)n);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  DRAG_CHECK(x && y && n >= 0, "drag_cast_f32_to_bf16: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, (long long)n);
  DRAG_LAUNCH_CHECK();
  return 0;
}
extern "C" int drag_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream) {
  DRAG_CHECK(x && y && n >= 0, "drag_cast_bf16_to_f32: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cast_b2f_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, (long long)n);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_timestep_embedding_bf16(const float* t, void* out, int32_t B, int32_t dim, void* stream) {
  DRAG_CHECK(t && out && B > 0 && dim > 0 && dim % 2 == 0, "drag_timestep_embedding_bf16: bad args");
  const int n = B * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t,
                     (bf16_t*)out, B, dim);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_patchify_u8(const void* img, void* out, int32_t B, int32_t H, int32_t W, int32_t P, int32_t ldo,
                                const float* mean3, const float* std3, void* stream) {
  DRAG_CHECK(img && out && mean3 && std3, "drag_patchify_u8: null pointer");
  // a VALID (unpadded) strided conv: trailing pixels that do not fill a patch are dropped (SigLIP 384/14 -> 27)
  DRAG_CHECK(B > 0 && P > 0 && H >= P && W >= P && ldo >= 3 * P * P, "drag_patchify_u8: bad shape");
  PatchArgs p;
  p.img = (const uint8_t*)img; p.out = (bf16_t*)out; p.B = B; p.H = H; p.W = W; p.P = P; p.ldo = ldo;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.std[c] = std3[c]; }
  const long long total = (long long)B * (H / P) * (W / P) * ldo;
  hipLaunchKernelGGL(patchify_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_scale_sum_bf16(const void* x, const float* scales, void* out, int32_t G, int32_t N, int64_t elems,
                                   void* stream) {
  DRAG_CHECK(x && scales && out && G > 0 && N > 0 && elems > 0, "drag_scale_sum_bf16: bad args");
  hipLaunchKernelGGL(scale_sum_kernel, dim3(ew_grid((long long)G * elems)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, scales, (bf16_t*)out, G, N, (long long)elems);
  DRAG_LAUNCH_CHECK();
  return 0;
}
