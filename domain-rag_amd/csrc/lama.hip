// lama.hip — the LaMa inpainting generator (big-lama FFCResNetGenerator) on gfx950, float32 like the reference.
//
// Replaces simple_lama_inpainting.SimpleLama.__call__ as lama_inpaint/lama_inpaint.py:172-215 drives it (the wheel and
// the TorchScript export it wraps are un-vendored): prepare_img_and_mask -> FFC ResNet generator -> blend -> uint8.
//
//   conv2d_f32_kernel    every convolution of the network (7x7 / 3x3 / 1x1, stride 1|2, reflect or zero padding, the
//                        stride-2 transposed 3x3 of the up-sampling path) as an NHWC implicit GEMM on the exact-f32 matrix
//                        core (v_mfma_f32_32x32x2_f32 = an fmaf chain): 64 pixels x 64 output channels per workgroup, 16
//                        input channels of one tap per step through double-buffered LDS; BatchNorm (eval) as a per-channel
//                        scale/shift, the FFC branch sum (addend), ReLU / sigmoid and the residual add in the epilogue.
//   rfft2 / irfft2       torch.fft.rfftn / irfftn(norm="ortho") of the FourierUnit by direct summation against
//                        float64-computed twiddle tables: map sizes are H/8 x W/8 of arbitrary (non power-of-two) size and the
//                        whole transform is <3 % of the network's FLOPs, so four small VALU kernels (r2c along W, c2c along
//                        H and back) with 4 outputs per thread are enough.  Channel order of the spectrum is the reference's
//                        (2c = real, 2c+1 = imaginary).
//   lama_prepare / blend /255, symmetric pad to a multiple of 8, mask > 0, img*(1-m) | m ; m*pred + (1-m)*img, *255, clip, truncate.
#include "drag_common.h"

namespace {

constexpr int CT_M = 64, CT_N = 64, CT_K = 16, CT_LD = CT_K + 4;

struct ConvK {
  drag_conv2d_f32_args a;
  long long npix;
};

__device__ __forceinline__ bool tap_coord(int o, int k, int n_in, int stride, int pad, int mode, int transposed, int& i) {
  if (transposed) {
    const int t = o + pad - k;
    if (t < 0 || (t % stride) != 0) return false;
    i = t / stride;
    return i < n_in;
  }
  i = o * stride - pad + k;
  if (mode == DRAG_PAD_REFLECT) {
    if (i < 0) i = -i;
    if (i >= n_in) i = 2 * n_in - 2 - i;
    return true;
  }
  return i >= 0 && i < n_in;
}

__global__ __launch_bounds__(256) void conv2d_f32_kernel(ConvK p) {
  __shared__ __attribute__((aligned(16))) float As[2][CT_M][CT_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][CT_N][CT_LD];
  const drag_conv2d_f32_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const long long m0 = (long long)blockIdx.x * CT_M;
  const int n0 = blockIdx.y * CT_N;
  // loader role: one float4 of A and one of B per step
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  const long long lm = m0 + lr;
  const bool m_ok = lm < p.npix;
  int ob = 0, oy = 0, ox = 0;
  if (m_ok) {
    ox = (int)(lm % a.Wo);
    const long long t = lm / a.Wo;
    oy = (int)(t % a.Ho);
    ob = (int)(t / a.Ho);
  }
  const bool n_ok = n0 + lr < a.Cout;
  const int taps = a.KH * a.KW;
  const int kchunks = (a.Cin + CT_K - 1) / CT_K;
  const int nsteps = taps * kchunks;
  const float* wrow = a.w + (long long)(n0 + (n_ok ? lr : 0)) * taps * a.Cin;

  f32x4_t ra, rb;
  int tap = 0, kc = 0;
  const float* arow = nullptr;
  bool a_ok = false;
  auto set_tap = [&]() {
    int iy = 0, ix = 0;
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    a_ok = m_ok && tap_coord(oy, ky, a.Hi, a.stride, a.pad, a.pad_mode, a.transposed, iy) &&
           tap_coord(ox, kx, a.Wi, a.stride, a.pad, a.pad_mode, a.transposed, ix);
    arow = a.x + (((long long)ob * a.Hi + iy) * a.Wi + ix) * a.ldx;
  };
  auto fetch = [&]() {
    const int c = kc * CT_K + lk;
    const bool c_ok = c < a.Cin;
    ra = (a_ok && c_ok) ? *(const f32x4_t*)(arow + c) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    rb = (n_ok && c_ok) ? *(const f32x4_t*)(wrow + (long long)tap * a.Cin + c) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  auto advance = [&]() {
    if (++kc == kchunks) {
      kc = 0;
      ++tap;
      if (tap < taps) set_tap();
    }
  };
  auto stash = [&](int buf) {
    *(f32x4_t*)&As[buf][lr][lk] = ra;
    *(f32x4_t*)&Bs[buf][lr][lk] = rb;
  };

  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  set_tap();
  fetch();
  stash(0);
  __syncthreads();
  const int fi = lane & 31, fk = (lane >> 5) * 8;
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < nsteps;
    if (more) {
      advance();
      fetch();
    }
    const f32x4_t a0 = *(const f32x4_t*)&As[buf][wm * 32 + fi][fk], a1 = *(const f32x4_t*)&As[buf][wm * 32 + fi][fk + 4];
    const f32x4_t b0 = *(const f32x4_t*)&Bs[buf][wn * 32 + fi][fk], b1 = *(const f32x4_t*)&Bs[buf][wn * 32 + fi][fk + 4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc, 0, 0, 0);
    if (more) stash(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane owns output channel co and 16 pixel rows of the wave's 32x32 block
  const int co = n0 + wn * 32 + (lane & 31);
  if (co >= a.Cout) return;
  const float sc = a.scale ? a.scale[co] : 1.0f, sh = a.shift ? a.shift[co] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long m = m0 + wm * 32 + (lane >> 5) * 4 + 8 * (r >> 2) + (r & 3);
    if (m >= p.npix) continue;
    float v = acc[r];
    if (a.addend) v += a.addend[m * a.ld_add + co];
    v = v * sc + sh;
    if (a.act == DRAG_CONV_ACT_RELU) v = fmaxf(v, 0.f);
    else if (a.act == DRAG_CONV_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    if (a.resid) v += a.resid[m * a.ld_res + co];
    a.y[m * a.ldy + co] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// direct-summation DFTs.  tw_n[j] = (cos, sin)(2 pi j / n), j in [0, n).  TO = outputs per thread along the axis.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TO = 4;

// real [B,H,W,ldx] (C channels) -> complex z [B,H,Wf,C] (float2), e^{-i}
__global__ __launch_bounds__(256) void dft_w_r2c_kernel(const float* __restrict__ x, f32x2_t* __restrict__ z, const f32x2_t* __restrict__ tw,
                                                        int W, int Wf, int C, int ldx) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int k0 = (blockIdx.y * 4 + threadIdx.y) * TO;
  const long long row = blockIdx.z;   // b*H + h
  if (c >= C || k0 >= Wf) return;
  float re[TO], im[TO];
  int idx[TO];
#pragma unroll
  for (int t = 0; t < TO; ++t) re[t] = im[t] = 0.f, idx[t] = 0;
  const float* xp = x + row * W * ldx + c;
  for (int w = 0; w < W; ++w) {
    const float v = xp[(long long)w * ldx];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const f32x2_t cs = tw[idx[t]];
      re[t] = fmaf(v, cs[0], re[t]);
      im[t] = fmaf(-v, cs[1], im[t]);
      idx[t] += (k0 + t) % W;
      if (idx[t] >= W) idx[t] -= W;
    }
  }
#pragma unroll
  for (int t = 0; t < TO; ++t)
    if (k0 + t < Wf) z[(row * Wf + k0 + t) * C + c] = (f32x2_t){re[t], im[t]};
}

// complex [B,H,Wf,C] -> complex [B,H,Wf,C] along H; SIGN = -1 forward (e^{-i}), +1 inverse; result * norm
template <int SIGN>
__global__ __launch_bounds__(256) void dft_h_c2c_kernel(const f32x2_t* __restrict__ z, f32x2_t* __restrict__ y, const f32x2_t* __restrict__ tw,
                                                        int H, int WfC, float norm) {
  const int n = blockIdx.x * 64 + threadIdx.x;   // (kw, c)
  const int k0 = (blockIdx.y * 4 + threadIdx.y) * TO;
  const long long b = blockIdx.z;
  if (n >= WfC || k0 >= H) return;
  float re[TO], im[TO];
  int idx[TO];
#pragma unroll
  for (int t = 0; t < TO; ++t) re[t] = im[t] = 0.f, idx[t] = 0;
  const f32x2_t* zp = z + b * H * WfC + n;
  for (int h = 0; h < H; ++h) {
    const f32x2_t v = zp[(long long)h * WfC];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const f32x2_t cs = tw[idx[t]];
      const float s = SIGN < 0 ? -cs[1] : cs[1];   // e^{SIGN i theta} = cos + i s
      re[t] = fmaf(v[0], cs[0], fmaf(-v[1], s, re[t]));
      im[t] = fmaf(v[0], s, fmaf(v[1], cs[0], im[t]));
      idx[t] += (k0 + t) % H;
      if (idx[t] >= H) idx[t] -= H;
    }
  }
#pragma unroll
  for (int t = 0; t < TO; ++t)
    if (k0 + t < H) y[(b * H + k0 + t) * WfC + n] = (f32x2_t){re[t] * norm, im[t] * norm};
}

// complex [B,H,Wf,C] -> real [B,H,W,ldy] (c2r along W: the imaginary parts of the DC and Nyquist bins do not contribute),
// result * norm (+ add[B,H,W,ld_add])
__global__ __launch_bounds__(256) void dft_w_c2r_kernel(const f32x2_t* __restrict__ z, float* __restrict__ y, const float* __restrict__ add,
                                                        const f32x2_t* __restrict__ tw, int W, int Wf, int C, int ldy, int ld_add, float norm) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int w0 = (blockIdx.y * 4 + threadIdx.y) * TO;
  const long long row = blockIdx.z;
  if (c >= C || w0 >= W) return;
  float acc[TO];
  int idx[TO];
  const f32x2_t* zp = z + row * Wf * C + c;
  const float dc = zp[0][0];
#pragma unroll
  for (int t = 0; t < TO; ++t) acc[t] = dc, idx[t] = (w0 + t) % W;
  const int nyq = (W & 1) ? -1 : W / 2;
  for (int k = 1; k < Wf; ++k) {
    const f32x2_t v = zp[(long long)k * C];
    const float wk = k == nyq ? 1.0f : 2.0f;
    const float vr = v[0] * wk, vi = k == nyq ? 0.0f : v[1] * wk;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const f32x2_t cs = tw[idx[t]];
      acc[t] = fmaf(vr, cs[0], fmaf(-vi, cs[1], acc[t]));
      idx[t] += (w0 + t) % W;
      if (idx[t] >= W) idx[t] -= W;
    }
  }
#pragma unroll
  for (int t = 0; t < TO; ++t)
    if (w0 + t < W) {
      const long long o = row * W + w0 + t;
      float v = acc[t] * norm;
      if (add) v += add[o * ld_add + c];
      y[o * ldy + c] = v;
    }
}

__device__ __forceinline__ int sym_index(int i, int n) {   // numpy.pad(mode="symmetric"): ... c b a | a b c | c b a ...
  const int per = 2 * n;
  int j = i % per;
  return j < n ? j : per - 1 - j;
}

__global__ void lama_prepare_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask, float* __restrict__ x,
                                    int H, int W, int Hp, int Wp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Hp * Wp) return;
  const int px = (int)(i % Wp), py = (int)(i / Wp);
  const int sy = sym_index(py, H), sx = sym_index(px, W);
  const long long s = (long long)sy * W + sx;
  const float m = mask[s] > 0 ? 1.0f : 0.0f;
  f32x4_t o;
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = ((float)img[s * 3 + c] / 255.0f) * (1.0f - m);
  o[3] = m;
  *(f32x4_t*)(x + i * 4) = o;
}

__global__ void lama_blend_kernel(const float* __restrict__ pred, int ld, const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                  uint8_t* __restrict__ out, int H, int W, int Hp, int Wp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Hp * Wp) return;
  const int px = (int)(i % Wp), py = (int)(i / Wp);
  const long long s = (long long)sym_index(py, H) * W + sym_index(px, W);
  const float m = mask[s] > 0 ? 1.0f : 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float im = (float)img[s * 3 + c] / 255.0f;
    // two separately rounded products and one add, like the tensor expression mask * predicted + (1 - mask) * image
    const float v = __fadd_rn(__fmul_rn(m, pred[i * ld + c]), __fmul_rn(1.0f - m, im));
    const float s255 = fminf(fmaxf(__fmul_rn(v, 255.0f), 0.0f), 255.0f);
    out[i * 3 + c] = (uint8_t)s255;   // astype(uint8) truncates
  }
}

}  // namespace

extern "C" int drag_conv2d_f32(const drag_conv2d_f32_args* a, void* stream) {
  DRAG_CHECK(a && a->x && a->w && a->y, "conv2d_f32: null pointer");
  DRAG_CHECK(a->B > 0 && a->Ho > 0 && a->Wo > 0 && a->Hi > 0 && a->Wi > 0 && a->Cin > 0 && a->Cout > 0, "conv2d_f32: empty shape");
  DRAG_CHECK(a->Cin % 4 == 0 && a->ldx % 4 == 0 && ((uintptr_t)a->x & 15) == 0 && ((uintptr_t)a->w & 15) == 0,
             "conv2d_f32: input channels and pixel stride must be multiples of 4 floats, pointers 16-byte aligned");
  DRAG_CHECK(a->ldx >= a->Cin && a->ldy >= a->Cout, "conv2d_f32: pixel stride smaller than the channel count");
  DRAG_CHECK(a->KH >= 1 && a->KW >= 1 && a->stride >= 1 && a->pad >= 0, "conv2d_f32: bad kernel geometry");
  DRAG_CHECK(a->pad_mode == DRAG_PAD_ZERO || a->pad_mode == DRAG_PAD_REFLECT, "conv2d_f32: unknown padding mode");
  DRAG_CHECK(!(a->transposed && a->pad_mode == DRAG_PAD_REFLECT), "conv2d_f32: a transposed convolution pads with zeros");
  DRAG_CHECK(a->pad_mode != DRAG_PAD_REFLECT || (a->pad < a->Hi && a->pad < a->Wi), "conv2d_f32: reflect padding needs pad < input size");
  if (a->transposed) {
    DRAG_CHECK(a->Ho <= (a->Hi - 1) * a->stride - 2 * a->pad + a->KH + a->stride - 1 && a->Wo <= (a->Wi - 1) * a->stride - 2 * a->pad + a->KW + a->stride - 1,
               "conv2d_f32: transposed output larger than output_padding < stride allows");
  } else {
    DRAG_CHECK(a->Ho == (a->Hi + 2 * a->pad - a->KH) / a->stride + 1 && a->Wo == (a->Wi + 2 * a->pad - a->KW) / a->stride + 1,
               "conv2d_f32: output size does not match the geometry");
  }
  DRAG_CHECK(!a->addend || a->ld_add >= a->Cout, "conv2d_f32: addend pixel stride");
  DRAG_CHECK(!a->resid || a->ld_res >= a->Cout, "conv2d_f32: residual pixel stride");
  ConvK k;
  k.a = *a;
  k.npix = (long long)a->B * a->Ho * a->Wo;
  dim3 grid((unsigned)((k.npix + CT_M - 1) / CT_M), (unsigned)((a->Cout + CT_N - 1) / CT_N));
  hipLaunchKernelGGL(conv2d_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, k);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_rfft2_f32(const float* x, float* tmp, float* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx,
                              const float* tw_w, const float* tw_h, void* stream) {
  DRAG_CHECK(x && tmp && y && tw_w && tw_h, "rfft2_f32: null pointer");
  DRAG_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && ldx >= C, "rfft2_f32: bad shape");
  DRAG_CHECK((long long)B * H <= 65535, "rfft2_f32: too many rows");
  const int Wf = W / 2 + 1;
  const dim3 blk(64, 4);
  hipLaunchKernelGGL(dft_w_r2c_kernel, dim3((C + 63) / 64, (Wf + 4 * TO - 1) / (4 * TO), B * H), blk, 0, (hipStream_t)stream, x,
                     (f32x2_t*)tmp, (const f32x2_t*)tw_w, W, Wf, C, ldx);
  DRAG_LAUNCH_CHECK();
  const float norm = (float)(1.0 / sqrt((double)H * (double)W));
  hipLaunchKernelGGL(dft_h_c2c_kernel<-1>, dim3((Wf * C + 63) / 64, (H + 4 * TO - 1) / (4 * TO), B), blk, 0, (hipStream_t)stream,
                     (const f32x2_t*)tmp, (f32x2_t*)y, (const f32x2_t*)tw_h, H, Wf * C, norm);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_irfft2_f32(const float* f, float* tmp, float* y, const float* add, int32_t B, int32_t H, int32_t W, int32_t C,
                               int32_t ldy, int32_t ld_add, const float* tw_w, const float* tw_h, void* stream) {
  DRAG_CHECK(f && tmp && y && tw_w && tw_h, "irfft2_f32: null pointer");
  DRAG_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && (!add || ld_add >= C), "irfft2_f32: bad shape");
  DRAG_CHECK((long long)B * H <= 65535, "irfft2_f32: too many rows");
  const int Wf = W / 2 + 1;
  const dim3 blk(64, 4);
  hipLaunchKernelGGL(dft_h_c2c_kernel<1>, dim3((Wf * C + 63) / 64, (H + 4 * TO - 1) / (4 * TO), B), blk, 0, (hipStream_t)stream,
                     (const f32x2_t*)f, (f32x2_t*)tmp, (const f32x2_t*)tw_h, H, Wf * C, 1.0f);
  DRAG_LAUNCH_CHECK();
  const float norm = (float)(1.0 / sqrt((double)H * (double)W));
  hipLaunchKernelGGL(dft_w_c2r_kernel, dim3((C + 63) / 64, (W + 4 * TO - 1) / (4 * TO), B * H), blk, 0, (hipStream_t)stream,
                     (const f32x2_t*)tmp, y, add, (const f32x2_t*)tw_w, W, Wf, C, ldy, ld_add, norm);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_lama_prepare_u8(const void* img, const void* mask, float* x, int32_t H, int32_t W, int32_t Hp, int32_t Wp, void* stream) {
  DRAG_CHECK(img && mask && x, "lama_prepare_u8: null pointer");
  DRAG_CHECK(H > 0 && W > 0 && Hp >= H && Wp >= W, "lama_prepare_u8: bad shape");
  const long long n = (long long)Hp * Wp;
  hipLaunchKernelGGL(lama_prepare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img,
                     (const uint8_t*)mask, x, H, W, Hp, Wp);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_lama_blend_u8(const float* pred, int32_t ld, const void* img, const void* mask, void* out, int32_t H, int32_t W,
                                  int32_t Hp, int32_t Wp, void* stream) {
  DRAG_CHECK(pred && img && mask && out, "lama_blend_u8: null pointer");
  DRAG_CHECK(H > 0 && W > 0 && Hp >= H && Wp >= W && ld >= 3, "lama_blend_u8: bad shape");
  const long long n = (long long)Hp * Wp;
  hipLaunchKernelGGL(lama_blend_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, ld, (const uint8_t*)img,
                     (const uint8_t*)mask, (uint8_t*)out, H, W, Hp, Wp);
  DRAG_LAUNCH_CHECK();
  return 0;
}
