// resample.hip — Pillow-exact 8-bit separable image resample on gfx950 (byte / int32 work, HBM-bound).
//
// Replaces PIL.Image.resize (libImaging ImagingResample, 8bpc paths) where the reference resizes every image that
// enters a vision tower: openai-CLIP `_transform` Resize(224, BICUBIC)+CenterCrop behind clip.load
// (retrieval/clip100_resnet_style_all_shots.py:209,171,270-287) and SiglipImageProcessor's 384x384 BICUBIC inside
// FluxPriorReduxPipeline (batch_generate_flux_kshot.py:459-465, outpainting_updown_sampling_redux.py:1237-1243).
// The host computes Pillow's 22-bit fixed-point coefficient tables (domain-rag_amd/resample.py); the kernels apply
// them exactly as Pillow does:  out = clamp8(((1 << 21) + sum_k pixel[first + k] * kk[k]) >> 22), horizontal pass
// first (rounded to uint8, only the rows the vertical pass reads), then vertical.  Results are bit-identical to PIL,
// so embeddings — and therefore top-k indices — do not depend on where the resize ran.
//
// Layout: interleaved HWC uint8 with explicit row / image strides (a crop is a pointer offset + sliced tables).
// Horizontal pass: one thread per output pixel (all channels), windows of neighbouring threads overlap -> L1 hits.
// Vertical pass: one thread per 4 consecutive output bytes of a row -> fully coalesced 4-byte loads per tap.
#include "drag_common.h"

namespace {

constexpr int RS_PREC = 32 - 8 - 2;   // Pillow PRECISION_BITS

__device__ __forceinline__ uint32_t clamp8(int v) { return (uint32_t)min(max(v >> RS_PREC, 0), 255); }

struct RsArgs {
  const uint8_t* src;
  uint8_t* dst;
  const int32_t* kk;      // [n_out, ksize]
  const int32_t* bounds;  // [n_out, 2] = (first source index, tap count)
  int ksize, n_out;       // outputs along the resampled axis
  int lines;              // horizontal: rows per image; vertical: bytes per output row (out_w * channels)
  int channels;
  int first_off;          // subtracted from bounds[.][0] (vertical pass over a row-window temp image)
  long long src_img, dst_img;
  int src_row, dst_row;   // strides in bytes
};

template <int C>
__global__ __launch_bounds__(256) void resample_h_kernel(RsArgs p) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z;
  if (x >= p.n_out) return;
  const int first = p.bounds[2 * x] - p.first_off, n = p.bounds[2 * x + 1];
  const int32_t* k = p.kk + (long long)x * p.ksize;
  const uint8_t* s = p.src + b * p.src_img + (long long)y * p.src_row + (long long)first * C;
  int acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 1 << (RS_PREC - 1);
  for (int i = 0; i < n; ++i) {
    const int w = k[i];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += (int)s[i * C + c] * w;
  }
  uint8_t* d = p.dst + b * p.dst_img + (long long)y * p.dst_row + (long long)x * C;
#pragma unroll
  for (int c = 0; c < C; ++c) d[c] = (uint8_t)clamp8(acc[c]);
}

// RGB fast path of the horizontal pass (channels == 3, ksize <= 16): the generic kernel above issues
// one byte load per tap and channel and is bound by vector-memory instruction issue (it measured 0.49 TB/s of
// algorithmic bytes).  Here a thread owns one output column for RS_ROWS consecutive rows: its <= 16 weights live in
// registers (the table rows are zero-padded to ksize), and a row's taps — 3*n contiguous bytes at an arbitrary
// alignment — are fetched as <= 13 aligned dwords, realigned with v_alignbyte and unpacked 4 pixels per 3 dwords.
// Dwords past the taps are multiplied by zero weights; their addresses are clamped into the source region (the dword
// holding the first tap may start up to 3 bytes before it: still inside the caller's 4-byte aligned allocation).
constexpr int RS_ROWS = 4;
__global__ __launch_bounds__(256) void resample_h_rgb_kernel(RsArgs p) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y0 = blockIdx.y * RS_ROWS, b = blockIdx.z;
  if (x >= p.n_out) return;
  const int first = p.bounds[2 * x] - p.first_off;
  int w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = i < p.ksize ? p.kk[(long long)x * p.ksize + i] : 0;
  // last dword that lies wholly inside the source region of this launch
  const uintptr_t aend = (uintptr_t)p.src + (long long)(gridDim.z - 1) * p.src_img + (long long)p.lines * p.src_row;
  const uintptr_t amax = (aend - 4) & ~(uintptr_t)3;
  for (int r = 0; r < RS_ROWS; ++r) {
    const int y = y0 + r;
    if (y >= p.lines) break;
    const uintptr_t addr = (uintptr_t)(p.src + b * p.src_img + (long long)y * p.src_row + (long long)first * 3);
    const uintptr_t base = addr & ~(uintptr_t)3;
    const unsigned sh = (unsigned)(addr & 3);
    uint32_t d[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const uintptr_t a = base + 4 * i;
      if (a <= amax) {
        d[i] = *(const uint32_t*)a;
      } else {                       // the region's last 1-3 bytes (and everything past it): byte loads, zero beyond the end
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (a + k < aend) v |= (uint32_t)(*(const uint8_t*)(a + k)) << (8 * k);
        d[i] = v;
      }
    }
    int a0 = 1 << (RS_PREC - 1), a1 = a0, a2 = a0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // stream dwords 3j .. 3j+2 start at byte o + 12 j: pixels 4j .. 4j+3
      const uint32_t s0 = __builtin_amdgcn_alignbyte(d[3 * j + 1], d[3 * j], sh);
      const uint32_t s1 = __builtin_amdgcn_alignbyte(d[3 * j + 2], d[3 * j + 1], sh);
      const uint32_t s2 = __builtin_amdgcn_alignbyte(d[3 * j + 3], d[3 * j + 2], sh);
      const int w0 = w[4 * j], w1 = w[4 * j + 1], w2 = w[4 * j + 2], w3 = w[4 * j + 3];
      a0 += (int)(s0 & 255u) * w0 + (int)(s0 >> 24) * w1 + (int)((s1 >> 16) & 255u) * w2 + (int)((s2 >> 8) & 255u) * w3;
      a1 += (int)((s0 >> 8) & 255u) * w0 + (int)(s1 & 255u) * w1 + (int)(s1 >> 24) * w2 + (int)((s2 >> 16) & 255u) * w3;
      a2 += (int)((s0 >> 16) & 255u) * w0 + (int)((s1 >> 8) & 255u) * w1 + (int)(s2 & 255u) * w2 + (int)(s2 >> 24) * w3;
    }
    uint8_t* dst = p.dst + b * p.dst_img + (long long)y * p.dst_row + (long long)x * 3;
    dst[0] = (uint8_t)clamp8(a0); dst[1] = (uint8_t)clamp8(a1); dst[2] = (uint8_t)clamp8(a2);
  }
}

__global__ __launch_bounds__(256) void resample_v_kernel(RsArgs p) {
  const int q = blockIdx.x * 256 + threadIdx.x;       // 4-byte group inside the output row
  const int y = blockIdx.y, b = blockIdx.z;
  const int x0 = q * 4;
  if (x0 >= p.lines) return;
  const int first = p.bounds[2 * y] - p.first_off, n = p.bounds[2 * y + 1];
  const int32_t* k = p.kk + (long long)y * p.ksize;
  const uint8_t* s = p.src + b * p.src_img + (long long)first * p.src_row + x0;
  uint8_t* d = p.dst + b * p.dst_img + (long long)y * p.dst_row + x0;
  const bool full = x0 + 4 <= p.lines && ((((uintptr_t)s | (uintptr_t)d) & 3) == 0) && (p.src_row & 3) == 0;
  int a0 = 1 << (RS_PREC - 1), a1 = a0, a2 = a0, a3 = a0;
  if (full) {
    for (int i = 0; i < n; ++i) {
      const int w = k[i];
      const uint32_t v = *(const uint32_t*)(s + (long long)i * p.src_row);
      a0 += (int)(v & 255u) * w; a1 += (int)((v >> 8) & 255u) * w; a2 += (int)((v >> 16) & 255u) * w; a3 += (int)(v >> 24) * w;
    }
    *(uint32_t*)d = clamp8(a0) | (clamp8(a1) << 8) | (clamp8(a2) << 16) | (clamp8(a3) << 24);
  } else {
    const int m = min(4, p.lines - x0);
    for (int j = 0; j < m; ++j) {
      int a = 1 << (RS_PREC - 1);
      for (int i = 0; i < n; ++i) a += (int)s[(long long)i * p.src_row + j] * k[i];
      d[j] = (uint8_t)clamp8(a);
    }
  }
}

__global__ __launch_bounds__(256) void copy_window_kernel(const uint8_t* src, uint8_t* dst, int row_bytes, int src_row, int dst_row,
                                                          long long src_img, long long dst_img) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= row_bytes) return;
  dst[blockIdx.z * dst_img + (long long)blockIdx.y * dst_row + x] = src[blockIdx.z * src_img + (long long)blockIdx.y * src_row + x];
}

}  // namespace

extern "C" int drag_resample_u8(const drag_resample_args* a, void* stream) {
  DRAG_CHECK(a != nullptr && a->src && a->dst, "drag_resample_u8: null pointer");
  DRAG_CHECK(a->batch > 0 && a->batch <= 65535 && a->channels >= 1 && a->channels <= 4, "drag_resample_u8: batch in [1, 65535], channels in [1, 4]");
  DRAG_CHECK(a->src_h > 0 && a->src_w > 0 && a->out_h > 0 && a->out_w > 0 && a->out_h <= 65535 && a->src_h <= 65535,
             "drag_resample_u8: sizes must be in [1, 65535]");
  DRAG_CHECK((a->kx == nullptr) == (a->bx == nullptr) && (a->ky == nullptr) == (a->by == nullptr),
             "drag_resample_u8: a pass needs both its weight and its bounds table");
  DRAG_CHECK(a->kx || a->src_col0 + a->out_w <= a->src_w, "drag_resample_u8: without a horizontal pass the window must lie inside the source");
  DRAG_CHECK(a->ky || a->src_row0 + a->out_h <= a->src_h, "drag_resample_u8: without a vertical pass the window must lie inside the source");
  DRAG_CHECK(!(a->kx && a->ky) || (a->tmp && a->tmp_rows > 0 && a->tmp_row0 >= 0 && a->tmp_row0 + a->tmp_rows <= a->src_h),
             "drag_resample_u8: both passes need tmp and a row window inside the source");
  DRAG_CHECK((!a->kx || a->ksize_x > 0) && (!a->ky || a->ksize_y > 0), "drag_resample_u8: ksize must be positive");
  const int C = a->channels;
  hipStream_t st = (hipStream_t)stream;
  RsArgs h{}, v{};
  const int tmp_row_bytes = a->out_w * C;
  if (a->kx) {
    // rows that go through the horizontal pass: the vertical pass's window, or exactly the output rows
    const int row0 = a->ky ? a->tmp_row0 : a->src_row0;
    const int rows = a->ky ? a->tmp_rows : a->out_h;
    h.src = a->src + (long long)row0 * a->src_row_stride;
    h.dst = a->ky ? a->tmp : a->dst;
    h.kk = a->kx; h.bounds = a->bx; h.ksize = a->ksize_x; h.n_out = a->out_w; h.lines = rows; h.channels = C; h.first_off = 0;
    h.src_img = a->src_image_stride; h.src_row = a->src_row_stride;
    h.dst_img = a->ky ? (long long)a->tmp_rows * tmp_row_bytes : a->dst_image_stride;
    h.dst_row = a->ky ? tmp_row_bytes : a->dst_row_stride;
    const dim3 grid((a->out_w + 255) / 256, rows, a->batch);
    const bool rgb_fast = C == 3 && a->ksize_x <= 16 && (long long)rows * a->src_row_stride >= 8;
    if (rgb_fast) {
      hipLaunchKernelGGL(resample_h_rgb_kernel, dim3((a->out_w + 255) / 256, (rows + RS_ROWS - 1) / RS_ROWS, a->batch), dim3(256), 0, st, h);
    } else
    switch (C) {
      case 1: hipLaunchKernelGGL(resample_h_kernel<1>, grid, dim3(256), 0, st, h); break;
      case 2: hipLaunchKernelGGL(resample_h_kernel<2>, grid, dim3(256), 0, st, h); break;
      case 3: hipLaunchKernelGGL(resample_h_kernel<3>, grid, dim3(256), 0, st, h); break;
      default: hipLaunchKernelGGL(resample_h_kernel<4>, grid, dim3(256), 0, st, h); break;
    }
    DRAG_LAUNCH_CHECK();
  }
  if (a->ky) {
    v.src = a->kx ? a->tmp : a->src + (long long)a->src_col0 * C;
    v.dst = a->dst;
    v.kk = a->ky; v.bounds = a->by; v.ksize = a->ksize_y; v.n_out = a->out_h; v.lines = tmp_row_bytes; v.channels = C;
    v.first_off = a->kx ? a->tmp_row0 : 0;
    v.src_img = a->kx ? (long long)a->tmp_rows * tmp_row_bytes : a->src_image_stride;
    v.src_row = a->kx ? tmp_row_bytes : a->src_row_stride;
    v.dst_img = a->dst_image_stride; v.dst_row = a->dst_row_stride;
    const dim3 grid(((tmp_row_bytes + 3) / 4 + 255) / 256, a->out_h, a->batch);
    hipLaunchKernelGGL(resample_v_kernel, grid, dim3(256), 0, st, v);
    DRAG_LAUNCH_CHECK();
  }
  if (!a->kx && !a->ky) {     // same size: Pillow returns a copy
    const dim3 grid((tmp_row_bytes + 255) / 256, a->out_h, a->batch);
    hipLaunchKernelGGL(copy_window_kernel, grid, dim3(256), 0, st,
                       a->src + (long long)a->src_row0 * a->src_row_stride + (long long)a->src_col0 * C, a->dst, tmp_row_bytes,
                       a->src_row_stride, a->dst_row_stride, (long long)a->src_image_stride, (long long)a->dst_image_stride);
    DRAG_LAUNCH_CHECK();
  }
  return 0;
}

// --------------------------------------------------------------------------------------------
// cv2.resize(img, (OW, OH)) with INTER_LINEAR for a BATCH of differently sized RGB images -> float32 NCHW in [0, 1]:
// the input of the ResNet-stem style vector (compute_resnet_features, retrieval/clip100_resnet_style_all_shots.py:186-196:
// imread -> cvtColor -> resize(256, 256) -> /255).  The arithmetic is OpenCV's 8-bit linear path as restated in
// retrieval.cv2_resize_linear_u8 (two taps per axis, 11-bit fixed-point weights, horizontal pass into int32, vertical pass
// ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2): the HOST builds the per-image tap / weight tables with that very numpy
// code, this kernel only applies them, so the two agree bit for bit.  One thread per output pixel (3 channels).
// --------------------------------------------------------------------------------------------
namespace {
struct CvResizeArgs {
  const uint8_t* src;        // blob of RGB images [h, w, 3]
  const long long* src_off;  // [n] byte offset of each image
  const int* hw;             // [n, 2] (h, w)
  const int* tab;            // [n, 7, OW|OH]: sx, sx1, a0, a1 (length OW) then y0, y1, b0|b1 packed ... see host
  float* dst;                // [n, 3, OH, OW]
  int n, OH, OW;
};

__global__ __launch_bounds__(256) void cv_resize_linear_kernel(CvResizeArgs p) {
  const int i = blockIdx.y;
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= p.OH * p.OW) return;
  const int y = px / p.OW, x = px - y * p.OW;
  const int w = p.hw[2 * i + 1];
  const int L = p.OW > p.OH ? p.OW : p.OH;
  const int* t = p.tab + (long long)i * 8 * L;
  const int sx = t[x], sx1 = t[L + x], a0 = t[2 * L + x], a1 = t[3 * L + x];
  const int y0 = t[4 * L + y], y1 = t[5 * L + y], b0 = t[6 * L + y], b1 = t[7 * L + y];
  const uint8_t* s = p.src + p.src_off[i];
  const uint8_t* r0 = s + (long long)y0 * w * 3;
  const uint8_t* r1 = s + (long long)y1 * w * 3;
  float* d = p.dst + (long long)i * 3 * p.OH * p.OW + px;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int S0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
    const int S1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
    int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : v > 255 ? 255 : v;
    d[(long long)c * p.OH * p.OW] = (float)v / 255.0f;
  }
}
}  // namespace

extern "C" int drag_cv_resize_linear_u8_f32(const void* src, const int64_t* src_off, const int32_t* hw, const int32_t* tab, float* dst,
                                            int32_t n, int32_t out_h, int32_t out_w, void* stream) {
  DRAG_CHECK(src && src_off && hw && tab && dst, "drag_cv_resize_linear_u8_f32: null pointer");
  DRAG_CHECK(n > 0 && n <= 65535 && out_h > 0 && out_w > 0, "drag_cv_resize_linear_u8_f32: bad shape");
  CvResizeArgs p;
  p.src = (const uint8_t*)src; p.src_off = (const long long*)src_off; p.hw = hw; p.tab = tab; p.dst = dst;
  p.n = n; p.OH = out_h; p.OW = out_w;
  hipLaunchKernelGGL(cv_resize_linear_kernel, dim3((out_h * out_w + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}
