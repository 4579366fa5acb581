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
//   rfft2 / irfft2       torch.fft.rfftn / irfftn(norm="ortho") of the FourierUnit: map sizes are H/8 x W/8 of arbitrary (non
//                        power-of-two) size, so each of the four passes (r2c along W, c2c along H and back) is a dense DFT
//                        as a GEMM on the same f32 matrix core, its twiddle tiles gathered from a float64-computed n-entry
//                        table (dft_mfma_kernel).  Channel order of the spectrum is the reference's (2c real, 2c+1 imaginary).
//   lama_prepare / blend /255, symmetric pad to a multiple of 8, mask > 0, img*(1-m) | m ; m*pred + (1-m)*img, *255, clip, truncate.
#include "drag_common.h"
#include <stdlib.h>

namespace {


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

// CT_K = input channels of one tap per step.  16 when many workgroups share a CU (they hide each other's load latency);
// 64 for the low-resolution maps of the FFC blocks, where a CU holds one workgroup and the one-step-ahead prefetch must
// cover the whole load latency by itself (4x the bytes in flight, 4x the MFMA work per barrier).
// TM = tile edge in units of 64: 64x64 (one 32x32 accumulator per wave) or, for launches large enough to still fill the chip,
// 128x128 (2x2 accumulators per wave: half the global / LDS bytes per flop, 4x the MFMA work per barrier).
// Every instantiation adds the products of one output element in the same order (taps outer, channels ascending): same bits.
// LIN: a 1x1, stride-1, unpadded convolution whose tiles are all interior and whose channel count is a multiple of the step (every
// Linear of the float32 CLIP tower at batch sizes that are multiples of 64): no tap arithmetic and no per-load predicates — the general
// form wraps each of its 8-16 global loads per step in an exec-mask branch (75 % matrix-pipe busy on (51200, 3072, 768)).  Same
// products in the same order: same bits.
template <int CT_K, int TM, bool LIN = false>
__global__ __launch_bounds__(256) void conv2d_f32_kernel(ConvK p) {
  constexpr int CT_LD = CT_K + 4, NV = CT_K / 16, TILE = 64 * TM;
  __shared__ __attribute__((aligned(16))) float As[2][TILE][CT_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][TILE][CT_LD];
  const drag_conv2d_f32_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const long long m0 = (long long)blockIdx.x * TILE;
  const int n0 = blockIdx.y * TILE;
  // loader role: TM float4 of A and TM of B per 16 channels (rows lr, lr + 64)
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  bool m_ok[TM], n_ok[TM];
  int ob[TM], oy[TM], ox[TM];
  const float* wrow[TM];
  const int taps = a.KH * a.KW;
#pragma unroll
  for (int u = 0; u < TM; ++u) {
    const long long lm = m0 + lr + 64 * u;
    m_ok[u] = lm < p.npix;
    ob[u] = oy[u] = ox[u] = 0;
    if (m_ok[u]) {
      ox[u] = (int)(lm % a.Wo);
      const long long t = lm / a.Wo;
      oy[u] = (int)(t % a.Ho);
      ob[u] = (int)(t / a.Ho);
    }
    n_ok[u] = n0 + lr + 64 * u < a.Cout;
    wrow[u] = a.w + (long long)(n_ok[u] ? n0 + lr + 64 * u : 0) * taps * a.Cin;
  }
  const int kchunks = (a.Cin + CT_K - 1) / CT_K;
  const int nsteps = taps * kchunks;

  f32x4_t ra[TM][NV], rb[TM][NV];
  int tap = 0, kc = 0;
  const float* arow[TM];
  bool a_ok[TM];
  auto set_tap = [&]() {
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
#pragma unroll
    for (int u = 0; u < TM; ++u) {
      int iy = 0, ix = 0;
      a_ok[u] = m_ok[u] && tap_coord(oy[u], ky, a.Hi, a.stride, a.pad, a.pad_mode, a.transposed, iy) &&
                tap_coord(ox[u], kx, a.Wi, a.stride, a.pad, a.pad_mode, a.transposed, ix);
      arow[u] = a.x + (((long long)ob[u] * a.Hi + iy) * a.Wi + ix) * a.ldx;
    }
  };
  auto fetch = [&]() {
#pragma unroll
    for (int u = 0; u < TM; ++u)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = kc * CT_K + v * 16 + lk;
        if constexpr (LIN) {
          ra[u][v] = *(const f32x4_t*)(arow[u] + c);
          rb[u][v] = *(const f32x4_t*)(wrow[u] + c);
          continue;
        }
        const bool c_ok = c < a.Cin;
        ra[u][v] = (a_ok[u] && c_ok) ? *(const f32x4_t*)(arow[u] + c) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        rb[u][v] = (n_ok[u] && c_ok) ? *(const f32x4_t*)(wrow[u] + (long long)tap * a.Cin + c) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
  };
  auto advance = [&]() {
    if constexpr (LIN) { ++kc; return; }
    if (++kc == kchunks) {
      kc = 0;
      ++tap;
      if (tap < taps) set_tap();
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < TM; ++u)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        *(f32x4_t*)&As[buf][lr + 64 * u][v * 16 + lk] = ra[u][v];
        *(f32x4_t*)&Bs[buf][lr + 64 * u][v * 16 + lk] = rb[u][v];
      }
  };

  f32x16_t acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
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
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      f32x4_t af[TM][2], bf[TM][2];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float* ap = &As[buf][(wm * TM + i) * 32 + fi][v * 16 + fk];
        const float* bp = &Bs[buf][(wn * TM + i) * 32 + fi][v * 16 + fk];
        af[i][0] = *(const f32x4_t*)ap; af[i][1] = *(const f32x4_t*)(ap + 4);
        bf[i][0] = *(const f32x4_t*)bp; bf[i][1] = *(const f32x4_t*)(bp + 4);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TM; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][h][j], bf[n][h][j], acc[i][n], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }

  // epilogue: per 32x32 block a lane owns one output channel and 16 pixel rows
#pragma unroll
  for (int n = 0; n < TM; ++n) {
    const int co = n0 + (wn * TM + n) * 32 + (lane & 31);
    if (co >= a.Cout) continue;
    const float sc = a.scale ? a.scale[co] : 1.0f, sh = a.shift ? a.shift[co] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long m = m0 + (wm * TM + i) * 32 + (lane >> 5) * 4 + 8 * (r >> 2) + (r & 3);
        if (m >= p.npix) continue;
        float v = acc[i][n][r];
        if (a.addend) v += a.addend[m * a.ld_add + co];
        v = v * sc + sh;
        if (a.act == DRAG_CONV_ACT_RELU) v = fmaxf(v, 0.f);
        else if (a.act == DRAG_CONV_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
        else if (a.act == DRAG_CONV_ACT_QUICK_GELU) v = v * (1.0f / (1.0f + expf(-1.702f * v)));
        if (a.resid) v += a.resid[m * a.ld_res + co];
        a.y[m * a.ldy + co] = v;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Convolutions with <= 4 output channels (the generator's last layer: 64 -> 3, 7x7).  On the matrix core their N dimension
// would be padded 3 -> 64 (95 % of the MFMA work wasted: 4.5 of the 25 ms of a 1024x768 frame); here one thread owns one
// output pixel and all its channels on the VALU: a (16+KH-1) x (16+KW-1) pixel patch of 16 input channels is staged in LDS
// per step (each input pixel is fetched ~1.9x instead of KH*KW times) next to the step's weights, which every lane reads
// at the same address (LDS broadcast).
// ---------------------------------------------------------------------------------------------------------------
constexpr int SC_T = 16, SC_KMAX = 7, SC_C = 16, SC_LD = SC_C + 4, SC_P = SC_T + SC_KMAX - 1;

template <int NCO>
__global__ __launch_bounds__(256) void conv_small_cout_f32_kernel(ConvK p) {
  __shared__ __attribute__((aligned(16))) float tile[SC_P][SC_P][SC_LD];
  __shared__ __attribute__((aligned(16))) float wl[NCO][SC_KMAX * SC_KMAX][SC_C];
  const drag_conv2d_f32_args& a = p.a;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int ox0 = blockIdx.x * SC_T, oy0 = blockIdx.y * SC_T, b = blockIdx.z;
  const int ph = SC_T + a.KH - 1, pw = SC_T + a.KW - 1;
  const float* xb = a.x + (long long)b * a.Hi * a.Wi * a.ldx;
  float acc[NCO];
#pragma unroll
  for (int co = 0; co < NCO; ++co) acc[co] = 0.f;
  const int taps = a.KH * a.KW;
  for (int c0 = 0; c0 < a.Cin; c0 += SC_C) {
    __syncthreads();
    for (int i = threadIdx.x; i < ph * pw * (SC_C / 4); i += 256) {
      const int c4 = i & 3, pix = i >> 2;
      const int py = pix / pw, px = pix - py * pw;
      int iy = oy0 + py - a.pad, ix = ox0 + px - a.pad;
      bool ok = true;
      if (a.pad_mode == DRAG_PAD_REFLECT) {
        // patch pixels beyond the last output row / column of a ragged block may reflect twice: clamp, they are never used
        if (iy < 0) iy = -iy;
        if (iy >= a.Hi) iy = 2 * a.Hi - 2 - iy;
        if (ix < 0) ix = -ix;
        if (ix >= a.Wi) ix = 2 * a.Wi - 2 - ix;
        iy = min(max(iy, 0), a.Hi - 1);
        ix = min(max(ix, 0), a.Wi - 1);
      } else {
        ok = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
      }
      const f32x4_t v = ok ? *(const f32x4_t*)(xb + ((long long)iy * a.Wi + ix) * a.ldx + c0 + c4 * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
      *(f32x4_t*)&tile[py][px][c4 * 4] = v;
    }
    for (int i = threadIdx.x; i < NCO * taps * (SC_C / 4); i += 256) {     // this chunk's weights: read back as LDS broadcasts
      const int c4 = i & 3, t = (i >> 2) % taps, co = (i >> 2) / taps;
      *(f32x4_t*)&wl[co][t][c4 * 4] = co < a.Cout ? *(const f32x4_t*)(a.w + ((long long)co * taps + t) * a.Cin + c0 + c4 * 4)
                                                   : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    for (int t = 0; t < taps; ++t) {
      const int ky = t / a.KW, kx = t - ky * a.KW;
#pragma unroll
      for (int c4 = 0; c4 < SC_C / 4; ++c4) {
        const f32x4_t v = *(const f32x4_t*)&tile[ty + ky][tx + kx][c4 * 4];
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
          const f32x4_t w4 = *(const f32x4_t*)&wl[co][t][c4 * 4];
          acc[co] = fmaf(v[0], w4[0], acc[co]);
          acc[co] = fmaf(v[1], w4[1], acc[co]);
          acc[co] = fmaf(v[2], w4[2], acc[co]);
          acc[co] = fmaf(v[3], w4[3], acc[co]);
        }
      }
    }
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy >= a.Ho || ox >= a.Wo) return;
  const long long m = ((long long)b * a.Ho + oy) * a.Wo + ox;
#pragma unroll
  for (int co = 0; co < NCO; ++co) {
    if (co >= a.Cout) break;
    float v = acc[co];
    if (a.addend) v += a.addend[m * a.ld_add + co];
    v = v * (a.scale ? a.scale[co] : 1.0f) + (a.shift ? a.shift[co] : 0.0f);
    if (a.act == DRAG_CONV_ACT_RELU) v = fmaxf(v, 0.f);
    else if (a.act == DRAG_CONV_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    else if (a.act == DRAG_CONV_ACT_QUICK_GELU) v = v * (1.0f / (1.0f + expf(-1.702f * v)));
    if (a.resid) v += a.resid[m * a.ld_res + co];
    a.y[m * a.ldy + co] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// DFT passes as f32-MFMA GEMMs against twiddle tiles generated on the fly.  tw[j] = (cos, sin)(2 pi j / n), j in [0, n),
// float64-evaluated by the caller; the element (i, k) of the n-point DFT matrix is tw[(i * k) mod n], so a thread that
// fills 4 consecutive k of row i walks the table with an integer stride — no n x n matrices in memory.
//
//   P[i, col] = sum_k  a_k cos(2 pi i k / n) * X[k, col]        Q[i, col] = sum_k  a_k sin(2 pi i k / n) * X[k, col]
//
// X[k, col] = x[z * x_zs + k * x_ks + col] with `col` running over contiguous floats (channels, or interleaved complex
// channels).  Per mode, with (re, im) = (even, odd) columns and `partner` the other column of the pair:
//   R2C  real columns:                 out(i, col) = (P, -Q)                       [r2c along W]
//   C2C  complex columns, sign s:      re = P_re + s Q_im,  im = P_im - s Q_re     [s = +1: e^{-i}, forward; -1 inverse]
//   C2R  complex columns, a_k = 1|2:   out(i, col/2) = P_re - Q_im (+ add)         [c2r along W: DC / Nyquist imaginary parts drop]
// Tile: 64 outputs x 64 columns per workgroup, 4 waves x (32x32 P and 32x32 Q accumulators), K-step 16.
// ---------------------------------------------------------------------------------------------------------------
enum { DFT_R2C = 0, DFT_C2C = 1, DFT_C2R = 2 };
constexpr int DT_M = 64, DT_N = 64, DT_K = 16, DT_LDA = DT_K + 4, DT_LDX = DT_N + 4;

struct DftK {
  const float* x;
  float* y;
  const float* add;
  const f32x2_t* tw;
  int n;            // transform length (period of the twiddle table)
  int M, K, N;      // outputs along the axis, contraction length, float columns
  long long x_zs, x_ks, y_zs, y_is, add_zs, add_is;
  float sgn, norm;
};

template <int MODE>
__global__ __launch_bounds__(256) void dft_mfma_kernel(DftK p) {
  __shared__ __attribute__((aligned(16))) float Cs[2][DT_M][DT_LDA];
  __shared__ __attribute__((aligned(16))) float Ss[2][DT_M][DT_LDA];
  __shared__ __attribute__((aligned(16))) float Xs[2][DT_K][DT_LDX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int i0 = blockIdx.y * DT_M, c0 = blockIdx.x * DT_N;
  const long long z = blockIdx.z;
  // twiddle loader role: row ti, 4 consecutive k
  const int ti = tid >> 2, tk = (tid & 3) * 4;
  const int gi = i0 + ti;
  const bool i_ok = gi < p.M;
  const int im = i_ok ? gi % p.n : 0;
  const int step16 = (int)(((long long)im * DT_K) % p.n);
  int idx0 = (int)(((long long)im * tk) % p.n);      // (i * k) mod n at k = k0 + tk
  // data loader role: row xk, 4 consecutive columns
  const int xk = tid >> 4, xc = (tid & 15) * 4;
  const bool c_ok = c0 + xc < p.N;
  const float* xp = p.x + z * p.x_zs + c0 + xc;
  const int nyq = (p.n & 1) ? -1 : p.n / 2;

  f32x4_t rc, rs, rx;
  auto fetch = [&](int k0) {
    int idx = idx0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + tk + j;
      f32x2_t t = (i_ok && k < p.K) ? p.tw[idx] : (f32x2_t){0.f, 0.f};
      if (MODE == DFT_C2R) {
        const bool edge = k == 0 || k == nyq;
        t[0] *= edge ? 1.0f : 2.0f;
        t[1] = edge ? 0.0f : 2.0f * t[1];
      }
      rc[j] = t[0];
      rs[j] = t[1];
      idx += im;
      if (idx >= p.n) idx -= p.n;
    }
    idx0 += step16;
    if (idx0 >= p.n) idx0 -= p.n;
    rx = (c_ok && k0 + xk < p.K) ? *(const f32x4_t*)(xp + (long long)(k0 + xk) * p.x_ks) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  auto stash = [&](int buf) {
    *(f32x4_t*)&Cs[buf][ti][tk] = rc;
    *(f32x4_t*)&Ss[buf][ti][tk] = rs;
    *(f32x4_t*)&Xs[buf][xk][xc] = rx;
  };
  f32x16_t P, Q;
#pragma unroll
  for (int r = 0; r < 16; ++r) P[r] = Q[r] = 0.f;
  const int nsteps = (p.K + DT_K - 1) / DT_K;
  fetch(0);
  stash(0);
  __syncthreads();
  const int fi = lane & 31, fk = (lane >> 5) * 8;
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < nsteps;
    if (more) fetch((s + 1) * DT_K);
    const f32x4_t a0 = *(const f32x4_t*)&Cs[buf][wm * 32 + fi][fk], a1 = *(const f32x4_t*)&Cs[buf][wm * 32 + fi][fk + 4];
    const f32x4_t s0 = *(const f32x4_t*)&Ss[buf][wm * 32 + fi][fk], s1 = *(const f32x4_t*)&Ss[buf][wm * 32 + fi][fk + 4];
    float xb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[j] = Xs[buf][fk + j][wn * 32 + fi];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      P = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], xb[j], P, 0, 0, 0);
      Q = __builtin_amdgcn_mfma_f32_32x32x2f32(s0[j], xb[j], Q, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      P = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], xb[4 + j], P, 0, 0, 0);
      Q = __builtin_amdgcn_mfma_f32_32x32x2f32(s1[j], xb[4 + j], Q, 0, 0, 0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
  // lane owns column col and 16 output rows; its pair partner (col ^ 1) sits in lane ^ 1 with the same rows
  const int col = c0 + wn * 32 + (lane & 31);
  const bool odd = lane & 1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + wm * 32 + (lane >> 5) * 4 + 8 * (r >> 2) + (r & 3);
    const float pq = MODE == DFT_R2C ? 0.f : __shfl_xor(Q[r], 1, 64);
    if (i >= p.M || col >= p.N) continue;
    if (MODE == DFT_R2C) {
      *(f32x2_t*)(p.y + z * p.y_zs + (long long)i * p.y_is + 2 * col) = (f32x2_t){P[r] * p.norm, -Q[r] * p.norm};
    } else if (MODE == DFT_C2C) {
      const float v = odd ? P[r] - p.sgn * pq : P[r] + p.sgn * pq;
      p.y[z * p.y_zs + (long long)i * p.y_is + col] = v * p.norm;
    } else {
      if (odd) continue;
      float v = (P[r] - pq) * p.norm;
      if (p.add) v += p.add[z * p.add_zs + (long long)i * p.add_is + (col >> 1)];
      p.y[z * p.y_zs + (long long)i * p.y_is + (col >> 1)] = v;
    }
  }
}

template <int MODE>
static int launch_dft(const DftK& k, long long Z, hipStream_t stream) {
  dim3 grid((unsigned)((k.N + DT_N - 1) / DT_N), (unsigned)((k.M + DT_M - 1) / DT_M), (unsigned)Z);
  hipLaunchKernelGGL(dft_mfma_kernel<MODE>, grid, dim3(256), 0, stream, k);
  DRAG_LAUNCH_CHECK();
  return 0;
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
  // ldx < Cin = "packed row" reads: a pixel's Cin floats run on into its right-hand neighbours (KW taps x channels of an NHWC
  // row are contiguous), which turns a KH x KW x C kernel into KH x 1 x (KW*C) with no padded channel groups; the caller
  // guarantees the last read of a row stays inside it
  const bool packed_row = a->ldx < a->Cin && a->KW == 1 && a->pad == 0 && !a->transposed &&
                          (long long)(a->Wo - 1) * a->stride * a->ldx + a->Cin <= (long long)a->Wi * a->ldx;
  DRAG_CHECK((a->ldx >= a->Cin || packed_row) && a->ldy >= a->Cout, "conv2d_f32: pixel stride smaller than the channel count");
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
  static const bool no_small = getenv("DRAG_CONV_NO_SMALL_COUT") != nullptr;
  if (a->Cout <= 4 && a->stride == 1 && !a->transposed && a->KH <= SC_KMAX && a->KW <= SC_KMAX && a->Cin % SC_C == 0 && a->ldx >= a->Cin &&
      a->B <= 65535 && !no_small) {
    dim3 grid((unsigned)((a->Wo + SC_T - 1) / SC_T), (unsigned)((a->Ho + SC_T - 1) / SC_T), (unsigned)a->B);
    if (a->Cout == 1)
      hipLaunchKernelGGL(conv_small_cout_f32_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, k);
    else
      hipLaunchKernelGGL(conv_small_cout_f32_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, k);
    DRAG_LAUNCH_CHECK();
    return 0;
  }
  // tile policy (speed only — every instantiation produces the same bits): 128x128 tiles when that still launches >= 2
  // workgroups per CU, else 64x64 with 16-channel steps, or 64-channel steps when even those leave CUs with one workgroup
  static const char* force = getenv("DRAG_CONV_TILE");            // "64x16" | "64x64" | "128" (tests, ablations)
  const long long g128 = ((k.npix + 127) / 128) * ((a->Cout + 127) / 128);
  const long long g64 = ((k.npix + 63) / 64) * ((a->Cout + 63) / 64);
  int pick = (a->Cout >= 128 && g128 >= 512) ? 2 : ((a->Cin % 64 == 0 && g64 < 1024) ? 1 : 0);
  if (force) pick = force[0] == '1' ? 2 : (force[3] == '6' ? 1 : 0);
  if (pick == 1 && a->Cin % 64 != 0) pick = 0;
  static const bool no_lin = getenv("DRAG_CONV_NO_LIN") != nullptr;
  const int tile = pick == 2 ? 128 : 64, step = pick == 1 ? 64 : 16;
  const bool lin = !no_lin && a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad == 0 && !a->transposed && a->Hi == a->Ho && a->Wi == a->Wo &&
                   k.npix % tile == 0 && a->Cout % tile == 0 && a->Cin % step == 0;
  if (pick == 2) {
    dim3 grid((unsigned)((k.npix + 127) / 128), (unsigned)((a->Cout + 127) / 128));
    if (lin) hipLaunchKernelGGL((conv2d_f32_kernel<16, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, k);
    else hipLaunchKernelGGL((conv2d_f32_kernel<16, 2>), grid, dim3(256), 0, (hipStream_t)stream, k);
  } else {
    dim3 grid((unsigned)((k.npix + 63) / 64), (unsigned)((a->Cout + 63) / 64));
    if (pick == 1) {
      if (lin) hipLaunchKernelGGL((conv2d_f32_kernel<64, 1, true>), grid, dim3(256), 0, (hipStream_t)stream, k);
      else hipLaunchKernelGGL((conv2d_f32_kernel<64, 1>), grid, dim3(256), 0, (hipStream_t)stream, k);
    } else {
      if (lin) hipLaunchKernelGGL((conv2d_f32_kernel<16, 1, true>), grid, dim3(256), 0, (hipStream_t)stream, k);
      else hipLaunchKernelGGL((conv2d_f32_kernel<16, 1>), grid, dim3(256), 0, (hipStream_t)stream, k);
    }
  }
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_rfft2_f32(const float* x, float* tmp, float* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx,
                              const float* tw_w, const float* tw_h, void* stream) {
  DRAG_CHECK(x && tmp && y && tw_w && tw_h, "rfft2_f32: null pointer");
  DRAG_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && ldx >= C, "rfft2_f32: bad shape");
  DRAG_CHECK(C % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)tmp & 15) == 0 && ((uintptr_t)y & 15) == 0,
             "rfft2_f32: channels and pixel stride must be multiples of 4 floats, pointers 16-byte aligned");
  DRAG_CHECK((long long)B * H <= 65535, "rfft2_f32: too many rows");
  const int Wf = W / 2 + 1;
  DftK k{};
  // r2c along W: z = (b, h); X[w, c]; out tmp[z, kw, c] complex
  k.x = x; k.y = tmp; k.add = nullptr; k.tw = (const f32x2_t*)tw_w; k.n = W;
  k.M = Wf; k.K = W; k.N = C;
  k.x_zs = (long long)W * ldx; k.x_ks = ldx; k.y_zs = (long long)Wf * 2 * C; k.y_is = 2 * C;
  k.sgn = 1.f; k.norm = 1.f;
  if (int rc = launch_dft<DFT_R2C>(k, (long long)B * H, (hipStream_t)stream)) return rc;
  // c2c along H, e^{-i}: z = b; X[h, (kw, c, ri)]
  k.x = tmp; k.y = y; k.tw = (const f32x2_t*)tw_h; k.n = H;
  k.M = H; k.K = H; k.N = Wf * 2 * C;
  k.x_zs = (long long)H * k.N; k.x_ks = k.N; k.y_zs = k.x_zs; k.y_is = k.N;
  k.sgn = 1.f; k.norm = (float)(1.0 / sqrt((double)H * (double)W));
  return launch_dft<DFT_C2C>(k, B, (hipStream_t)stream);
}

extern "C" int drag_irfft2_f32(const float* f, float* tmp, float* y, const float* add, int32_t B, int32_t H, int32_t W, int32_t C,
                               int32_t ldy, int32_t ld_add, const float* tw_w, const float* tw_h, void* stream) {
  DRAG_CHECK(f && tmp && y && tw_w && tw_h, "irfft2_f32: null pointer");
  DRAG_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && (!add || ld_add >= C), "irfft2_f32: bad shape");
  DRAG_CHECK(C % 4 == 0 && ((uintptr_t)f & 15) == 0 && ((uintptr_t)tmp & 15) == 0,
             "irfft2_f32: channels must be a multiple of 4 floats, spectrum pointers 16-byte aligned");
  DRAG_CHECK((long long)B * H <= 65535, "irfft2_f32: too many rows");
  const int Wf = W / 2 + 1;
  DftK k{};
  k.x = f; k.y = tmp; k.add = nullptr; k.tw = (const f32x2_t*)tw_h; k.n = H;
  k.M = H; k.K = H; k.N = Wf * 2 * C;
  k.x_zs = (long long)H * k.N; k.x_ks = k.N; k.y_zs = k.x_zs; k.y_is = k.N;
  k.sgn = -1.f; k.norm = 1.f;
  if (int rc = launch_dft<DFT_C2C>(k, B, (hipStream_t)stream)) return rc;
  // c2r along W: z = (b, h); X[kw, (c, ri)]; out y[z, w, c]
  k.x = tmp; k.y = y; k.add = add; k.tw = (const f32x2_t*)tw_w; k.n = W;
  k.M = W; k.K = Wf; k.N = 2 * C;
  k.x_zs = (long long)Wf * 2 * C; k.x_ks = 2 * C; k.y_zs = (long long)W * ldy; k.y_is = ldy;
  k.add_zs = (long long)W * ld_add; k.add_is = ld_add;
  k.sgn = 1.f; k.norm = (float)(1.0 / sqrt((double)H * (double)W));
  return launch_dft<DFT_C2R>(k, (long long)B * H, (hipStream_t)stream);
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
