// png.hip — uint8 images in HBM -> complete PNG files in HBM (gfx950), the functions of png_core.h run in parallel.
// Replaces the reference's host-side `image.save(path)` of its large RGB results (batch_generate_flux_kshot.py:480,
// outpainting_updown_sampling_redux.py:1262,1278); SURVEY §8(f)-2.  A batch of n same-size images is encoded by seven
// stream-ordered launches, no host round trip in between; the caller reads the n file sizes and copies the bytes out.
//
//   1 png_filter_kernel   one workgroup per image row: the five PNG filters' costs (sum of |signed byte|, libpng's heuristic),
//                         the cheapest one applied, bytes counted into the image's histogram (LDS, then global atomics)
//   2 png_codes_kernel    one wave per image: symbols rank-sorted by the 64 lanes; the two-queue Huffman tree, canonical codes and
//                         the fixed-size block header by lane 0 (a few thousand serial steps); counts halved and redone if deeper than 15
//   3 png_bits_kernel     one thread per 64-byte chunk: its coded length in bits, and its Adler-32 partial sums
//   4 png_scan_kernel     exclusive scan of the chunk lengths (one workgroup per image)
//   5 png_pack_kernel     one thread per chunk: codes OR-ed into a 64-bit window, whole words stored, the two boundary words atomicOr-ed
//   6 png_assemble_kernel file bytes (signature, IHDR, IDAT header, zlib header, deflate bytes, Adler-32) and the CRC-32 of every
//                         256-byte piece of the IDAT chunk
//   7 png_finish_kernel   CRC pieces combined in GF(2) (crc(A||B) = crc(A) x^(8|B|) + crc(B)), IDAT length + CRC, IEND, file size
// All of it is integer work on ~3 bytes per pixel: HBM/LDS-bound, nothing for the matrix cores.
#include "drag_common.h"
#include "png_core.h"

namespace {

constexpr int CHUNK = 64;          // bytes of filtered data per thread in the bit-length / pack kernels
constexpr int PIECE = 256;         // bytes per CRC piece

struct PngArgs {
  const uint8_t* img;              // [n, H, W, C]
  int n, H, W, C;
  long long rb, nf;                // bytes per image row, filtered bytes per image = H * (1 + rb)
  long long nchunks, npieces_max;
  // per-image workspace regions (strides in elements of the region's type)
  uint8_t* filt; long long filt_stride;
  uint32_t* hist;                  // [n, 260]
  uint32_t* code;                  // [n, 260]  (length << 16) | reversed code
  uint32_t* chunk_bits; long long cb_stride;     // [n, nchunks]: bits per chunk, then (after the scan) exclusive offsets
  unsigned long long* sums;        // [n, 4]: Adler A sum, Adler B sum, total bits, unused
  uint32_t* words; long long words_stride;       // [n, ...] the deflate bit stream, LSB-first little-endian words
  uint32_t* piece_crc; long long pc_stride;      // [n, npieces_max]
  uint8_t* out; long long out_stride;
  long long* sizes;
  uint8_t ihdr[25];
};

__global__ __launch_bounds__(256) void png_filter_kernel(PngArgs p) {
  __shared__ uint32_t hist[256];
  __shared__ long long part[4][5];
  __shared__ int best_s;
  const int r = blockIdx.x, i = blockIdx.y, t = threadIdx.x;
  const uint8_t* cur = p.img + ((long long)i * p.H + r) * p.rb;
  const uint8_t* prev = cur - p.rb;
  const bool top = r == 0;
  const int C = p.C;
  hist[t] = 0;
  long long cost[5] = {0, 0, 0, 0, 0};
  for (long long x = t; x < p.rb; x += 256) {
    const int v = cur[x];
    const int a = x >= C ? cur[x - C] : 0, b = top ? 0 : prev[x], c = (top || x < C) ? 0 : prev[x - C];
#pragma unroll
    for (int f = 0; f < 5; ++f) cost[f] += png_filter_cost(png_filter_byte(f, v, a, b, c));
  }
#pragma unroll
  for (int f = 0; f < 5; ++f) {
    long long v = cost[f];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane_id() == 0) part[wave_id()][f] = v;
  }
  __syncthreads();
  if (t == 0) {
    int best = 0;
    long long bc = part[0][0] + part[1][0] + part[2][0] + part[3][0];
    for (int f = 1; f < 5; ++f) {
      const long long v = part[0][f] + part[1][f] + part[2][f] + part[3][f];
      if (v < bc) { bc = v; best = f; }          // ties: the lowest filter number
    }
    best_s = best;
  }
  __syncthreads();
  const int f = best_s;
  uint8_t* o = p.filt + (long long)i * p.filt_stride + (long long)r * (1 + p.rb);
  if (t == 0) { o[0] = (uint8_t)f; atomicAdd(&hist[f], 1u); }
  for (long long x = t; x < p.rb; x += 256) {
    const int v = cur[x];
    const int a = x >= C ? cur[x - C] : 0, b = top ? 0 : prev[x], c = (top || x < C) ? 0 : prev[x - C];
    const uint8_t y = png_filter_byte(f, v, a, b, c);
    o[1 + x] = y;
    atomicAdd(&hist[y], 1u);
  }
  __syncthreads();
  if (hist[t]) atomicAdd(&p.hist[(long long)i * 260 + t], hist[t]);
}

__global__ __launch_bounds__(64) void png_codes_kernel(PngArgs p) {
  __shared__ int32_t freq[520], par[520], sym[260], cnt[260];
  __shared__ uint8_t len[260];
  __shared__ uint32_t code[260];
  __shared__ uint32_t hdr[36];
  __shared__ int n_s, maxd_s;
  const int i = blockIdx.x, l = threadIdx.x;
  for (int s = l; s < 260; s += 64) {
    cnt[s] = s < 256 ? png_clamp_count(p.hist[(long long)i * 260 + s]) : (s == 256 ? 1 : 0);
    len[s] = 0;
  }
  if (l < 36) hdr[l] = 0;
  __syncthreads();
  for (;;) {
    // rank sort by (count, symbol): the order png_sort_symbols produces, found by all lanes
    if (l == 0) { int n = 0; for (int s = 0; s < PNG_NSYM; ++s) n += cnt[s] > 0; n_s = n; }
    for (int s = l; s < PNG_NSYM; s += 64) {
      const int cs = cnt[s];
      if (cs <= 0) continue;
      int rank = 0;
      for (int t = 0; t < PNG_NSYM; ++t) {
        const int ct = cnt[t];
        rank += ct > 0 && (ct < cs || (ct == cs && t < s));
      }
      freq[rank] = cs; sym[rank] = s;
    }
    __syncthreads();
    if (l == 0) maxd_s = png_tree_lengths(n_s, freq, par, sym, len);
    __syncthreads();
    if (maxd_s <= PNG_MAXBITS) break;
    for (int s = l; s < PNG_NSYM; s += 64)
      if (cnt[s] > 0) cnt[s] = (cnt[s] + 1) >> 1;
    __syncthreads();
  }
  if (l == 0) {
    png_canonical_codes(len, code);
    png_block_header(len, hdr);
  }
  __syncthreads();
  for (int s = l; s < PNG_NSYM; s += 64) p.code[(long long)i * 260 + s] = code[s];
  // header words: 34 whole ones and the 18 low bits of word 34, which the first data chunk ORs into afterwards (stream order)
  if (l < 35) p.words[(long long)i * p.words_stride + l] = hdr[l];
}

__global__ __launch_bounds__(256) void png_bits_kernel(PngArgs p) {
  __shared__ uint8_t len[260];
  __shared__ unsigned long long red[4][2];
  const int i = blockIdx.y, t = threadIdx.x;
  for (int s = t; s < PNG_NSYM; s += 256) len[s] = (uint8_t)(p.code[(long long)i * 260 + s] >> 16);
  __syncthreads();
  const long long c = (long long)blockIdx.x * 256 + t;
  unsigned long long sa = 0, sb = 0;
  if (c < p.nchunks) {
    const long long j0 = c * CHUNK;
    const uint8_t* d = p.filt + (long long)i * p.filt_stride + j0;
    const int m = (int)min((long long)CHUNK, p.nf - j0);
    uint32_t bits = 0;
    uint32_t ksum = 0;
    if (m == CHUNK) {
#pragma unroll
      for (int q = 0; q < CHUNK / 16; ++q) {
        const u32x4_t v = *(const u32x4_t*)(d + 16 * q);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const uint32_t byte = (v[k >> 2] >> (8 * (k & 3))) & 0xff;
          bits += len[byte]; sa += byte; ksum += (uint32_t)(16 * q + k) * byte;
        }
      }
    } else {
      for (int k = 0; k < m; ++k) { const uint32_t byte = d[k]; bits += len[byte]; sa += byte; ksum += (uint32_t)k * byte; }
    }
    if (c == p.nchunks - 1) bits += len[256];                 // the end-of-block code follows the last byte
    p.chunk_bits[(long long)i * p.cb_stride + c] = bits;
    // sum over the chunk of (nf - j) d_j = (nf - j0) * sa - sum k d_k
    sb = (unsigned long long)(p.nf - j0) * sa - ksum;
  }
#pragma unroll
  for (int m2 = 32; m2 >= 1; m2 >>= 1) { sa += __shfl_xor(sa, m2, 64); sb += __shfl_xor(sb, m2, 64); }
  if (lane_id() == 0) { red[wave_id()][0] = sa; red[wave_id()][1] = sb; }
  __syncthreads();
  if (t == 0) {
    atomicAdd(&p.sums[(long long)i * 4 + 0], red[0][0] + red[1][0] + red[2][0] + red[3][0]);
    atomicAdd(&p.sums[(long long)i * 4 + 1], red[0][1] + red[1][1] + red[2][1] + red[3][1]);
  }
}

__global__ __launch_bounds__(1024) void png_scan_kernel(PngArgs p) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const int i = blockIdx.x, t = threadIdx.x;
  uint32_t* a = p.chunk_bits + (long long)i * p.cb_stride;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (long long base = 0; base < p.nchunks; base += 1024) {
    const long long c = base + t;
    const uint32_t v = c < p.nchunks ? a[c] : 0u;
    uint32_t x = v;                                            // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d, 64);
      if (lane_id() >= d) x += y;
    }
    if (lane_id() == 63) wsum[wave_id()] = x;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wave_id(); ++w) off += wsum[w];
    if (c < p.nchunks) a[c] = off + x - v;                     // exclusive
    __syncthreads();
    if (t == 1023) carry_s = off + x;
    __syncthreads();
  }
  if (t == 0) p.sums[(long long)i * 4 + 2] = (unsigned long long)PNG_HEADER_BITS + carry_s;
}

__global__ __launch_bounds__(256) void png_pack_kernel(PngArgs p) {
  __shared__ uint32_t code[260];
  const int i = blockIdx.y, t = threadIdx.x;
  for (int s = t; s < PNG_NSYM; s += 256) code[s] = p.code[(long long)i * 260 + s];
  __syncthreads();
  const long long c = (long long)blockIdx.x * 256 + t;
  if (c >= p.nchunks) return;
  const long long j0 = c * CHUNK;
  const uint8_t* d = p.filt + (long long)i * p.filt_stride + j0;
  const int m = (int)min((long long)CHUNK, p.nf - j0);
  uint32_t* words = p.words + (long long)i * p.words_stride;
  const uint32_t off = PNG_HEADER_BITS + p.chunk_bits[(long long)i * p.cb_stride + c];
  uint32_t w = off >> 5;
  int nb = off & 31;
  unsigned long long acc = 0;
  bool first = true;
  auto put = [&](uint32_t cd) {
    acc |= (unsigned long long)(cd & 0xffffu) << nb;
    nb += (int)(cd >> 16);
    if (nb >= 32) {
      // the first word is shared with the previous chunk's last bits; later whole words belong to this chunk alone
      if (first) { atomicOr(&words[w], (uint32_t)acc); first = false; } else words[w] = (uint32_t)acc;
      ++w; acc >>= 32; nb -= 32;
    }
  };
  if (m == CHUNK) {
#pragma unroll
    for (int q = 0; q < CHUNK / 16; ++q) {
      const u32x4_t v = *(const u32x4_t*)(d + 16 * q);
#pragma unroll
      for (int k = 0; k < 16; ++k) put(code[(v[k >> 2] >> (8 * (k & 3))) & 0xff]);
    }
  } else {
    for (int k = 0; k < m; ++k) put(code[d[k]]);
  }
  if (c == p.nchunks - 1) put(code[256]);
  if (nb > 0) atomicOr(&words[w], (uint32_t)acc);
}

// byte q of the IDAT chunk's CRC domain (type + data) of image i
__device__ __forceinline__ uint8_t idat_byte(const PngArgs& p, int i, long long q, long long D, uint32_t adler) {
  if (q < 4) return (uint8_t)("IDAT"[q]);
  if (q == 4) return 0x78;
  if (q == 5) return 0x01;
  q -= 6;
  if (q < D) {
    const uint32_t w = p.words[(long long)i * p.words_stride + (q >> 2)];
    return (uint8_t)(w >> (8 * (q & 3)));
  }
  q -= D;
  return (uint8_t)(adler >> (8 * (3 - q)));
}

__device__ __forceinline__ void stream_geometry(const PngArgs& p, int i, long long& D, uint32_t& adler) {
  const unsigned long long bits = p.sums[(long long)i * 4 + 2];
  D = (long long)((bits + 7) >> 3);
  const unsigned long long A = (1 + p.sums[(long long)i * 4 + 0]) % 65521ull;
  const unsigned long long B = ((unsigned long long)p.nf + p.sums[(long long)i * 4 + 1]) % 65521ull;
  adler = (uint32_t)((B << 16) | A);
}

__global__ __launch_bounds__(256) void png_assemble_kernel(PngArgs p) {
  __shared__ uint32_t tab[256];
  const int i = blockIdx.y, t = threadIdx.x;
  tab[t] = png_crc_table_entry((uint32_t)t);
  __syncthreads();
  long long D; uint32_t adler;
  stream_geometry(p, i, D, adler);
  const long long cn = 4 + 2 + D + 4;                          // type + zlib stream
  uint8_t* out = p.out + (long long)i * p.out_stride;
  const long long piece = (long long)blockIdx.x * 256 + t;
  if (piece == 0) {
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    for (int k = 0; k < 8; ++k) out[k] = sig[k];
    for (int k = 0; k < 25; ++k) out[8 + k] = p.ihdr[k];
  }
  const long long q0 = piece * PIECE;
  if (q0 >= cn) return;
  const int m = (int)min((long long)PIECE, cn - q0);
  uint32_t crc = 0xffffffffu;
  for (int k = 0; k < m; ++k) {
    const uint8_t b = idat_byte(p, i, q0 + k, D, adler);
    out[37 + q0 + k] = b;                                      // the chunk type starts at byte 37 (after the 4-byte length)
    crc = tab[(crc ^ b) & 0xff] ^ (crc >> 8);
  }
  p.piece_crc[(long long)i * p.pc_stride + piece] = crc ^ 0xffffffffu;
}

__global__ __launch_bounds__(1024) void png_finish_kernel(PngArgs p) {
  __shared__ uint32_t crc_s[1024];
  __shared__ unsigned long long len_s[1024];
  const int i = blockIdx.x, t = threadIdx.x;
  long long D; uint32_t adler;
  stream_geometry(p, i, D, adler);
  const long long Z = 2 + D + 4, cn = 4 + Z;
  const long long np = (cn + PIECE - 1) / PIECE;
  const long long run = (np + 1023) / 1024;
  const uint32_t xp = png_x_pow_8n(PIECE);                     // every piece but the last is PIECE bytes long
  const long long a = t * run, b = min(np, a + run);
  uint32_t crc = 0; unsigned long long len = 0;
  for (long long k = a; k < b; ++k) {
    const uint32_t ck = p.piece_crc[(long long)i * p.pc_stride + k];
    const unsigned long long lk = (unsigned long long)min((long long)PIECE, cn - k * PIECE);
    crc = len == 0 ? ck : (lk == PIECE ? png_gf2_mulmod(xp, crc) ^ ck : png_crc_combine(crc, ck, lk));
    len += lk;
  }
  crc_s[t] = crc; len_s[t] = len;
  __syncthreads();
  for (int s = 1; s < 1024; s <<= 1) {
    if ((t & (2 * s - 1)) == 0) {
      const unsigned long long lb = len_s[t + s];
      if (lb) { crc_s[t] = len_s[t] ? png_crc_combine(crc_s[t], crc_s[t + s], lb) : crc_s[t + s]; len_s[t] += lb; }
    }
    __syncthreads();
  }
  if (t == 0) {
    uint8_t* out = p.out + (long long)i * p.out_stride;
    out[33] = (uint8_t)(Z >> 24); out[34] = (uint8_t)(Z >> 16); out[35] = (uint8_t)(Z >> 8); out[36] = (uint8_t)Z;
    uint8_t* q = out + PNG_FILE_PREFIX + Z;
    const uint32_t c = crc_s[0];
    q[0] = (uint8_t)(c >> 24); q[1] = (uint8_t)(c >> 16); q[2] = (uint8_t)(c >> 8); q[3] = (uint8_t)c;
    const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xae, 0x42, 0x60, 0x82};
    for (int k = 0; k < 12; ++k) q[4 + k] = iend[k];
    p.sizes[i] = PNG_FILE_PREFIX + Z + 4 + 12;
  }
}

inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

struct PngPlan {
  long long rb, nf, nchunks, max_d, npieces_max;
  long long filt_stride, cb_stride, words_stride, pc_stride;
  long long off_filt, off_hist, off_code, off_cb, off_sums, off_words, off_pc, total;
  long long out_stride;
};

bool make_plan(int n, int H, int W, int C, PngPlan& q) {
  if (n <= 0 || H <= 0 || W <= 0 || (C != 1 && C != 3)) return false;
  q.rb = (long long)W * C;
  q.nf = (long long)H * (1 + q.rb);
  if (q.nf >= (1ll << 26) || n > 65535 || H > 65535) return false;       // Adler partial sums and 32-bit bit offsets stay exact below 64 MiB per image
  q.nchunks = (q.nf + CHUNK - 1) / CHUNK;
  q.max_d = (PNG_HEADER_BITS + (q.nf + 1) * PNG_MAXBITS + 7) / 8;         // every code at most 15 bits
  q.npieces_max = (4 + 2 + q.max_d + 4 + PIECE - 1) / PIECE;
  q.filt_stride = align_up(q.nf, 64) + 64;
  q.cb_stride = align_up(q.nchunks, 64);
  q.words_stride = align_up(q.max_d / 4 + 4, 64);
  q.pc_stride = align_up(q.npieces_max, 64);
  long long o = 0;
  q.off_filt = o; o = align_up(o + n * q.filt_stride, 256);
  q.off_hist = o; o = align_up(o + (long long)n * 260 * 4, 256);
  q.off_code = o; o = align_up(o + (long long)n * 260 * 4, 256);
  q.off_cb = o; o = align_up(o + n * q.cb_stride * 4, 256);
  q.off_sums = o; o = align_up(o + (long long)n * 4 * 8, 256);
  q.off_words = o; o = align_up(o + n * q.words_stride * 4, 256);
  q.off_pc = o; o = align_up(o + n * q.pc_stride * 4, 256);
  q.total = o;
  q.out_stride = align_up(PNG_FILE_PREFIX + 2 + q.max_d + 4 + 4 + 12, 256);
  return true;
}

}  // namespace

// workspace / output sizing for a batch of n images [H, W, C]: *workspace_bytes for the scratch buffer, *out_stride bytes per
// image in the output buffer (the worst case: every code 15 bits long)
extern "C" int drag_png_plan(int32_t n, int32_t H, int32_t W, int32_t C, int64_t* workspace_bytes, int64_t* out_stride) {
  PngPlan q;
  DRAG_CHECK(make_plan(n, H, W, C, q), "drag_png_plan: n, H, W must be positive, channels 1 or 3, fewer than 2^26 filtered bytes per image");
  if (workspace_bytes) *workspace_bytes = q.total;
  if (out_stride) *out_stride = q.out_stride;
  return 0;
}

// images uint8 [n, H, W, C] (C = 1 grey, 3 RGB; dense) -> n PNG files at out + i * out_stride, their byte counts in sizes[i]
// (device int64).  workspace: drag_png_plan's size, 256-byte aligned.  Everything is enqueued on `stream`.
extern "C" int drag_png_encode(const void* images, int32_t n, int32_t H, int32_t W, int32_t C, void* workspace, int64_t workspace_bytes,
                               void* out, int64_t out_stride, int64_t* sizes, void* stream) {
  DRAG_CHECK(images && workspace && out && sizes, "drag_png_encode: null pointer");
  PngPlan q;
  DRAG_CHECK(make_plan(n, H, W, C, q), "drag_png_encode: n, H, W must be positive, channels 1 or 3, fewer than 2^26 filtered bytes per image");
  DRAG_CHECK(workspace_bytes >= q.total && out_stride >= q.out_stride, "drag_png_encode: workspace or out_stride smaller than drag_png_plan's");
  DRAG_CHECK(((uintptr_t)workspace & 255) == 0, "drag_png_encode: workspace must be 256-byte aligned");
  char* ws = (char*)workspace;
  PngArgs p;
  p.img = (const uint8_t*)images; p.n = n; p.H = H; p.W = W; p.C = C; p.rb = q.rb; p.nf = q.nf;
  p.nchunks = q.nchunks; p.npieces_max = q.npieces_max;
  p.filt = (uint8_t*)(ws + q.off_filt); p.filt_stride = q.filt_stride;
  p.hist = (uint32_t*)(ws + q.off_hist); p.code = (uint32_t*)(ws + q.off_code);
  p.chunk_bits = (uint32_t*)(ws + q.off_cb); p.cb_stride = q.cb_stride;
  p.sums = (unsigned long long*)(ws + q.off_sums);
  p.words = (uint32_t*)(ws + q.off_words); p.words_stride = q.words_stride;
  p.piece_crc = (uint32_t*)(ws + q.off_pc); p.pc_stride = q.pc_stride;
  p.out = (uint8_t*)out; p.out_stride = out_stride; p.sizes = (long long*)sizes;
  png_ihdr_chunk(W, H, C, p.ihdr);
  const hipStream_t st = (hipStream_t)stream;
  // zero: histograms .. sums (contiguous regions hist, code, chunk_bits, sums) and the bit stream (OR-ed into)
  DRAG_CHECK(hipMemsetAsync(ws + q.off_hist, 0, (size_t)(q.off_words - q.off_hist), st) == hipSuccess, "drag_png_encode: memset failed");
  DRAG_CHECK(hipMemsetAsync(ws + q.off_words, 0, (size_t)(q.off_pc - q.off_words), st) == hipSuccess, "drag_png_encode: memset failed");
  const dim3 gchunks((unsigned)((q.nchunks + 255) / 256), (unsigned)n);
  const dim3 gpieces((unsigned)((q.npieces_max + 255) / 256), (unsigned)n);
  hipLaunchKernelGGL(png_filter_kernel, dim3((unsigned)H, (unsigned)n), dim3(256), 0, st, p);
  hipLaunchKernelGGL(png_codes_kernel, dim3((unsigned)n), dim3(64), 0, st, p);
  hipLaunchKernelGGL(png_bits_kernel, gchunks, dim3(256), 0, st, p);
  hipLaunchKernelGGL(png_scan_kernel, dim3((unsigned)n), dim3(1024), 0, st, p);
  hipLaunchKernelGGL(png_pack_kernel, gchunks, dim3(256), 0, st, p);
  hipLaunchKernelGGL(png_assemble_kernel, gpieces, dim3(256), 0, st, p);
  hipLaunchKernelGGL(png_finish_kernel, dim3((unsigned)n), dim3(1024), 0, st, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}
