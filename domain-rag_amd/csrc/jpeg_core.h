// jpeg_core.h — baseline JPEG decoding arithmetic, byte-identical to libjpeg-turbo's default decompression (what
// PIL's Image.open(path).convert("RGB") returns for the corpus images the reference embeds one by one,
// retrieval/clip100_resnet_style_all_shots.py:270-281): Huffman entropy decoding, dequantisation + the "ISLOW" integer
// IDCT (jidctint.c), "fancy" (triangle) chroma upsampling (jdsample.c) and the fixed-point YCbCr -> RGB tables (jdcolor.c).
// The reference reaches libjpeg only through Pillow; the algorithms restated here are libjpeg's published ones.
//
// Host/device neutral on purpose: csrc/jpeg.hip wraps these functions in gfx950 kernels (one image per LANE for the
// sequential parts, one thread per 8x8 block / per pixel for the parallel ones), and tests/helpers/jpeg_host.cpp compiles the
// very same functions with g++ so that the arithmetic is checked against PIL on the build host, where there is no GPU.
// Nothing on the product path uses the host build.
//
// Supported: SOF0 / SOF1 (sequential Huffman, 8-bit), one interleaved scan, and (round 3) SOF2 PROGRESSIVE Huffman files (any
// number of scans: DC / AC, first / refinement, interleaved DC scans, table redefinitions between scans; jdphuff.c restated in
// jpeg_decode_progressive below); 1 component (-> grey replicated to RGB) or 3 components YCbCr with the luma at full
// resolution and the chroma at 1x1, 2x1 or 2x2 subsampling, restart intervals.
// Everything else (arithmetic, lossless, 12-bit, CMYK / RGB-coded, multi-scan sequential, other sampling ratios) is reported in
// JpegInfo::status and left to the caller; a progressive file whose scans stop before coefficients 0-9 of every component are
// fully refined is reported after decoding (libjpeg would run its inter-block smoothing on such a file).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define JHD __host__ __device__ __forceinline__
#else
#define JHD inline
#endif

enum {
  JPEG_OK = 0,
  JPEG_ERR_NOT_JPEG = 1,
  JPEG_ERR_TRUNCATED = 2,
  JPEG_ERR_PROGRESSIVE = 3,      // the non-Huffman / lossless / hierarchical frame types (SOF3, SOF5-7, SOF9-15); SOF2 itself is decoded
  JPEG_ERR_PRECISION = 4,        // not 8 bits per sample
  JPEG_ERR_COMPONENTS = 5,       // not 1 or 3 components, or a colour space other than grey / YCbCr
  JPEG_ERR_SAMPLING = 6,         // not 4:4:4 / 4:2:2 / 4:2:0 (full-resolution luma, chroma 1x1 / 2x1 / 2x2 subsampled)
  JPEG_ERR_MULTISCAN = 7,        // a scan that does not carry all components, or spectral selection / successive approximation
  JPEG_ERR_TABLES = 8,           // a referenced Huffman / quantisation table was never defined
  JPEG_ERR_TOO_SMALL = 9,        // subsampled chroma at most 2 samples wide (libjpeg switches to box replication there)
  JPEG_ERR_TOO_LARGE = 11,       // more than 2^24 pixels (a header can claim 65535 x 65535; a batch path plans memory per file: the caller decodes giants singly)
};

struct JpegInfo {               // == drag_jpeg_info (include/domainrag_hip.h): 48 x int32
  int32_t status;
  int32_t width, height, ncomp;
  int32_t hs[3], vs[3];         // sampling factors
  int32_t tq[3], td[3], ta[3];  // quantisation / DC / AC table ids per component
  int32_t hmax, vmax;
  int32_t mcus_x, mcus_y;
  int32_t restart_interval;
  int32_t scan_off;             // offset of the first entropy-coded byte
  int32_t dqt_off[4];           // offset of a table's first element (zigzag order), -1 = absent
  int32_t dqt_16[4];            // 1 = 16-bit elements
  int32_t dht_off[8];           // [class * 4 + id]: offset of the 16 code-length counts, -1 = absent
  int32_t progressive;          // 1 = SOF2: scan_off is the offset of the first SOS MARKER (its 0xFF), td / ta are per scan
  int32_t cid[3];               // component identifiers (scan headers name their components by these)
  int32_t reserved[3];
};

JHD int jpeg_natural_order(int k) {   // zigzag position -> natural (row-major) position; positions past 63 alias 63 like libjpeg
  const uint8_t t[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  return k < 64 ? t[k] : 63;
}

JHD int jpeg_u16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// ---------------------------------------------------------------------------------------------- header
// Walks the marker segments of one file.  Later DQT / DHT definitions override earlier ones (single scan: all of them
// precede SOS).  Never reads past `len`.
JHD void jpeg_parse(const uint8_t* d, int64_t len, JpegInfo* o) {
  for (int i = 0; i < (int)(sizeof(JpegInfo) / 4); ++i) ((int32_t*)o)[i] = 0;
  for (int i = 0; i < 4; ++i) o->dqt_off[i] = -1;
  for (int i = 0; i < 8; ++i) o->dht_off[i] = -1;
  if (len < 4 || d[0] != 0xFF || d[1] != 0xD8) { o->status = JPEG_ERR_NOT_JPEG; return; }
  int64_t p = 2;
  bool have_sof = false, jfif = false, adobe = false;
  int adobe_transform = 0;
  int cid[3] = {0, 0, 0};
  for (;;) {
    if (p + 4 > len) { o->status = JPEG_ERR_TRUNCATED; return; }
    if (d[p] != 0xFF) { o->status = JPEG_ERR_NOT_JPEG; return; }
    while (p < len && d[p] == 0xFF) ++p;                 // fill bytes
    if (p >= len) { o->status = JPEG_ERR_TRUNCATED; return; }
    const int m = d[p++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;    // parameterless
    if (m == 0xD9) { o->status = JPEG_ERR_TRUNCATED; return; }             // EOI before any scan
    if (p + 2 > len) { o->status = JPEG_ERR_TRUNCATED; return; }
    const int L = jpeg_u16(d + p);
    if (L < 2 || p + L > len) { o->status = JPEG_ERR_TRUNCATED; return; }
    const uint8_t* s = d + p + 2;
    const int n = L - 2;
    if (m == 0xE0) {
      if (n >= 5 && s[0] == 'J' && s[1] == 'F' && s[2] == 'I' && s[3] == 'F' && s[4] == 0) jfif = true;
    } else if (m == 0xEE) {
      if (n >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') { adobe = true; adobe_transform = s[11]; }
    } else if (m == 0xDB) {
      int q = 0;
      while (q < n) {
        const int pq = s[q] >> 4, id = s[q] & 15;
        const int sz = pq ? 128 : 64;
        if (id > 3 || q + 1 + sz > n) { o->status = JPEG_ERR_TABLES; return; }
        o->dqt_off[id] = (int32_t)(p + 2 + q + 1);
        o->dqt_16[id] = pq ? 1 : 0;
        q += 1 + sz;
      }
    } else if (m == 0xC4) {
      int q = 0;
      while (q < n) {
        if (q + 17 > n) { o->status = JPEG_ERR_TABLES; return; }
        const int tc = s[q] >> 4, id = s[q] & 15;
        int cnt = 0;
        for (int i = 0; i < 16; ++i) cnt += s[q + 1 + i];
        if (tc > 1 || id > 3 || cnt > 256 || q + 17 + cnt > n) { o->status = JPEG_ERR_TABLES; return; }
        o->dht_off[tc * 4 + id] = (int32_t)(p + 2 + q + 1);
        q += 17 + cnt;
      }
    } else if (m == 0xDD) {
      if (n >= 2) o->restart_interval = jpeg_u16(s);
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      o->progressive = m == 0xC2 ? 1 : 0;
      if (n < 6) { o->status = JPEG_ERR_TRUNCATED; return; }
      if (s[0] != 8) { o->status = JPEG_ERR_PRECISION; return; }
      o->height = jpeg_u16(s + 1); o->width = jpeg_u16(s + 3); o->ncomp = s[5];
      if (o->ncomp != 1 && o->ncomp != 3) { o->status = JPEG_ERR_COMPONENTS; return; }
      if (n < 6 + 3 * o->ncomp || o->width <= 0 || o->height <= 0) { o->status = JPEG_ERR_TRUNCATED; return; }
      for (int c = 0; c < o->ncomp; ++c) {
        cid[c] = s[6 + 3 * c]; o->cid[c] = cid[c];
        o->hs[c] = s[7 + 3 * c] >> 4; o->vs[c] = s[7 + 3 * c] & 15; o->tq[c] = s[8 + 3 * c];
        if (o->tq[c] > 3) { o->status = JPEG_ERR_TABLES; return; }
      }
      have_sof = true;
    } else if ((m >= 0xC3 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      o->status = JPEG_ERR_PROGRESSIVE; return;
    } else if (m == 0xDA) {
      if (!have_sof) { o->status = JPEG_ERR_NOT_JPEG; return; }
      if (o->progressive) { o->scan_off = (int32_t)(p - 2); break; }     // the scans are walked by jpeg_decode_progressive
      if (n < 1 || s[0] != o->ncomp || n < 1 + 2 * o->ncomp + 3) { o->status = JPEG_ERR_MULTISCAN; return; }
      for (int c = 0; c < o->ncomp; ++c) {
        if (s[1 + 2 * c] != cid[c]) { o->status = JPEG_ERR_MULTISCAN; return; }
        o->td[c] = s[2 + 2 * c] >> 4; o->ta[c] = s[2 + 2 * c] & 15;
        if (o->td[c] > 3 || o->ta[c] > 3) { o->status = JPEG_ERR_TABLES; return; }
      }
      const uint8_t* t = s + 1 + 2 * o->ncomp;
      if (t[0] != 0 || t[1] != 63 || t[2] != 0) { o->status = JPEG_ERR_MULTISCAN; return; }
      o->scan_off = (int32_t)(p + L);
      break;
    }
    p += L;
  }
  // ---- what the pixel pipeline below can do
  if (o->ncomp == 1) {
    o->hs[0] = o->vs[0] = 1;                      // a single-component scan is never interleaved: 1x1 MCUs whatever SOF says
  } else {
    // libjpeg's colour-space guess (jdapimin.c default_decompress_parms): JFIF -> YCbCr; Adobe transform 1 -> YCbCr, 0 -> RGB;
    // neither: component ids 'R','G','B' -> RGB, anything else -> YCbCr
    bool ycc = true;
    if (jfif) ycc = true;
    else if (adobe) ycc = adobe_transform == 1;
    else if (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B') ycc = false;
    if (!ycc) { o->status = JPEG_ERR_COMPONENTS; return; }
    // 4:4:4, 4:2:2 (h2v1), 4:2:0 (h2v2).  4:4:0 (h1v2) has a fancy upsampler in libjpeg-turbo too, but no encoder at hand
    // writes it, so its arithmetic could not be checked against the library: left to the caller rather than guessed.
    const int h0 = o->hs[0], v0 = o->vs[0];
    if (!((h0 == 1 && v0 == 1) || (h0 == 2 && v0 == 1) || (h0 == 2 && v0 == 2)) || o->hs[1] != 1 || o->vs[1] != 1 || o->hs[2] != 1 ||
        o->vs[2] != 1) {
      o->status = JPEG_ERR_SAMPLING; return;
    }
  }
  if ((int64_t)o->width * o->height > ((int64_t)1 << 24)) { o->status = JPEG_ERR_TOO_LARGE; return; }
  o->hmax = o->hs[0]; o->vmax = o->vs[0];
  o->mcus_x = (o->width + 8 * o->hmax - 1) / (8 * o->hmax);
  o->mcus_y = (o->height + 8 * o->vmax - 1) / (8 * o->vmax);
  for (int c = 0; c < o->ncomp; ++c) {
    if (o->dqt_off[o->tq[c]] < 0) { o->status = JPEG_ERR_TABLES; return; }
    if (!o->progressive && (o->dht_off[o->td[c]] < 0 || o->dht_off[4 + o->ta[c]] < 0)) { o->status = JPEG_ERR_TABLES; return; }
  }
  // jdsample.c picks the triangle filters only when downsampled_width > 2 (box replication otherwise): images under 5 pixels wide
  if (o->ncomp == 3 && o->hmax == 2 && (o->width + 1) / 2 <= 2) { o->status = JPEG_ERR_TOO_SMALL; return; }
}

// geometry of component c's sample plane as the IDCT writes it (whole blocks)
JHD int jpeg_blocks_w(const JpegInfo* o, int c) { return o->mcus_x * o->hs[c]; }
JHD int jpeg_blocks_h(const JpegInfo* o, int c) { return o->mcus_y * o->vs[c]; }
JHD int64_t jpeg_total_blocks(const JpegInfo* o) {
  int64_t t = 0;
  for (int c = 0; c < o->ncomp; ++c) t += (int64_t)jpeg_blocks_w(o, c) * jpeg_blocks_h(o, c);
  return t;
}

// ---------------------------------------------------------------------------------------------- Huffman tables
// A decoding table is seen through a VIEW type T (plain arrays on the host, lane-interleaved LDS on the GPU):
//   T::LB          lookahead bits of the direct table
//   t.lut(i)       uint16 [1 << LB]: (code length << 8) | symbol for codes of <= LB bits, 0 otherwise
//   t.limk(l)      uint32, l = 1..16: low half = exclusive upper bound, left-aligned to 16 bits, of the codes of length <= l
//                  (canonical codes grow with their length, so these bounds are monotonic and a 16-bit window w belongs to
//                  the smallest l with w < bound); high half = index of the first symbol of length l
//   t.val(i)       uint8 [T::NV]: the symbols in code order (symbols past NV are dropped: the caller rejects such tables)
// Every per-symbol access therefore stays in fast memory; nothing walks maxcode[] bit by bit (libjpeg's slow path does, from
// cached memory — on a GPU lane that walk is a chain of dependent memory round trips).
template <typename T>
JHD void jpeg_build_huff(const uint8_t* counts /* 16 counts, then the symbols */, T t) {
  for (int i = 0; i < (1 << T::LB); ++i) t.lut(i) = 0;
  const uint8_t* vals = counts + 16;
  int code = 0, k = 0;
  unsigned bound = 0;
  for (int l = 1; l <= 16; ++l) {
    const int n = counts[l - 1];
    const int kfirst = k;
    for (int i = 0; i < n; ++i, ++k, ++code) {
      if (k < T::NV) t.val(k) = vals[k];
      if (l <= T::LB) {
        const int base = code << (T::LB - l);
        for (int f = 0; f < (1 << (T::LB - l)); ++f) t.lut((base + f) & ((1 << T::LB) - 1)) = (uint16_t)((l << 8) | vals[k]);
      }
    }
    if (n) {
      const unsigned b = (unsigned)code << (16 - l);        // first window value NOT belonging to a code of length <= l
      bound = b > 0xFFFFu ? 0xFFFFu : b;
    }
    t.limk(l) = (uint32_t)bound | ((uint32_t)(kfirst & 0xFFFF) << 16);
    code <<= 1;
  }
}

// ---------------------------------------------------------------------------------------------- bit reader
struct JpegBits {
  const uint8_t* d;
  int64_t pos, end;
  uint64_t buf;       // MSB-aligned
  int cnt;
  int marker;         // a marker was met: the reader feeds zero bits from there on (libjpeg's behaviour at end of data)
  // read-ahead window: three aligned 8-byte words of the stream held in registers.  `cur` covers [wpos, wpos + 8), `nxt` the
  // next 8 bytes, `pre` the 8 after those — `pre` was requested one window shift (>= 8 consumed bytes, i.e. ~10 symbols)
  // before anything looks at it, so its memory latency is off the per-symbol critical path.  With 64 independent streams per
  // wave some lane misses the cache on almost every request: a load-then-use reader pays a full miss latency per refill.
  int64_t wpos;
  uint64_t cur, nxt, pre;
};

JHD uint64_t jpeg_load8_aligned(const uint8_t* p) {      // p is 8-byte aligned; first stream byte in bits 0-7
  uint64_t w;
  __builtin_memcpy(&w, __builtin_assume_aligned(p, 8), 8);
  return w;
}

// Every file of a batch must be followed by >= JPEG_TAIL_PAD readable bytes (the next files, or the blob's padding), and the
// blob must start on an 8-byte boundary: the window reads aligned words around the position without looking at `end`.
#define JPEG_TAIL_PAD 32

JHD void jpeg_bits_init(JpegBits* b, const uint8_t* d, int64_t pos, int64_t end) {
  b->d = d; b->pos = pos; b->end = end; b->buf = 0; b->cnt = 0; b->marker = 0;
  const int64_t a = (int64_t)((uintptr_t)(d + pos) & 7);
  b->wpos = pos - a;
  b->cur = jpeg_load8_aligned(d + b->wpos);
  b->nxt = jpeg_load8_aligned(d + b->wpos + 8);
  b->pre = jpeg_load8_aligned(d + b->wpos + 16);
}

// the 8 stream bytes at `pos` out of the register window, first byte in bits 0-7; shifts the window when pos has left `cur`
JHD uint64_t jpeg_window8(JpegBits* b) {
  while (b->pos - b->wpos >= 8) {                 // at most twice (a restart may jump further: re-seed instead)
    if (b->pos - b->wpos >= 24) {
      const int64_t a = (int64_t)((uintptr_t)(b->d + b->pos) & 7);
      b->wpos = b->pos - a;
      b->cur = jpeg_load8_aligned(b->d + b->wpos);
      b->nxt = jpeg_load8_aligned(b->d + b->wpos + 8);
      b->pre = jpeg_load8_aligned(b->d + b->wpos + 16);
      break;
    }
    b->cur = b->nxt; b->nxt = b->pre; b->wpos += 8;
    b->pre = jpeg_load8_aligned(b->d + b->wpos + 16);
  }
  const unsigned o = (unsigned)(b->pos - b->wpos) * 8;
  return o ? (b->cur >> o) | (b->nxt << (64 - o)) : b->cur;
}

// Guarantees > 32 valid bits (one code of <= 16 bits plus <= 16 extra bits) whenever the stream has them.  The bytes come out
// of the register window: four at once when none of them is 0xFF (entropy-coded data is 0xFF-free almost everywhere), byte by
// byte otherwise (stuffed zeros, markers) — no memory access on either path.  The last 8 bytes of the file go through the
// byte-wise memory loop below.
JHD void jpeg_bits_fill(JpegBits* b) {
  if (b->cnt > 32) return;
  if (!b->marker && b->pos + 8 <= b->end) {
    const uint64_t w = jpeg_window8(b);
    const uint32_t lo = (uint32_t)w;
    if ((((~lo) - 0x01010101u) & lo & 0x80808080u) == 0) {          // none of the first four bytes is 0xFF
      b->buf |= (uint64_t)__builtin_bswap32(lo) << (32 - b->cnt);  // the stream is big-endian
      b->cnt += 32;
      b->pos += 4;
      return;
    }
    int used = 0;
    while (b->cnt <= 56 && used < 7) {            // byte `used + 1` of the window is always available to look at
      unsigned byte = (unsigned)(w >> (8 * used)) & 0xFF;
      if (byte == 0xFF) {
        const unsigned nx = (unsigned)(w >> (8 * used + 8)) & 0xFF;
        if (nx == 0) used += 2;                   // stuffed zero
        else { b->marker = (int)nx; break; }      // leave pos at the 0xFF
      } else {
        used += 1;
      }
      b->buf |= (uint64_t)byte << (56 - b->cnt);
      b->cnt += 8;
    }
    b->pos += used;
    if (b->cnt > 32 || !b->marker) return;        // (at least 4 bytes were consumed unless a marker stopped the loop)
  }
  while (b->cnt <= 56) {
    unsigned byte = 0;
    if (!b->marker && b->pos < b->end) {
      byte = b->d[b->pos];
      if (byte == 0xFF) {
        const unsigned nx = b->pos + 1 < b->end ? b->d[b->pos + 1] : 0xD9;
        if (nx == 0) b->pos += 2;                      // stuffed zero
        else { b->marker = (int)nx; byte = 0; }        // leave pos at the 0xFF
      } else {
        b->pos += 1;
      }
    } else if (!b->marker) {
      b->marker = 0xD9;
    }
    b->buf |= (uint64_t)byte << (56 - b->cnt);
    b->cnt += 8;
  }
}
JHD int jpeg_bits_peek(const JpegBits* b, int n) { return (int)(b->buf >> (64 - n)); }
JHD void jpeg_bits_skip(JpegBits* b, int n) { b->buf <<= n; b->cnt -= n; }
JHD int jpeg_receive_extend(JpegBits* b, int s) {       // s in 1..16
  const int r = jpeg_bits_peek(b, s);
  jpeg_bits_skip(b, s);
  return r < (1 << (s - 1)) ? r - (1 << s) + 1 : r;
}

// one symbol.  The long-code data (16 packed words) is fetched unconditionally next to the direct-table entry: the loads
// are independent of each other, so a symbol costs one fast-memory latency (two for a code longer than LB bits).
template <typename T>
JHD int jpeg_decode_symbol(JpegBits* b, T t) {
  const unsigned w16 = (unsigned)jpeg_bits_peek(b, 16);
  const unsigned e = t.lut((int)(w16 >> (16 - T::LB)));
  uint32_t lk[17];
#pragma unroll
  for (int l = T::LB + 1; l <= 16; ++l) lk[l] = t.limk(l);
  const uint32_t lk_lb = t.limk(T::LB);
  if (e) { jpeg_bits_skip(b, (int)(e >> 8)); return (int)(e & 255); }
  int len = 17;
  uint32_t prev_bound = lk_lb & 0xFFFFu, kfirst = 0, first = 0;
#pragma unroll
  for (int l = 16; l > T::LB; --l) {              // smallest l with w16 < bound(l): scan downwards, keep the last hit
    const uint32_t bound = lk[l] & 0xFFFFu;
    const uint32_t below = l - 1 > T::LB ? (lk[l - 1] & 0xFFFFu) : prev_bound;
    if (w16 < bound) { len = l; kfirst = lk[l] >> 16; first = below; }
  }
  if (len > 16) { jpeg_bits_skip(b, 16); return 0; }     // not a code of this table (corrupt data): libjpeg warns and returns 0
  jpeg_bits_skip(b, len);
  return t.val((int)((kfirst + ((w16 - first) >> (16 - len))) & (T::NV - 1)));
}

// after an MCU count hits the restart interval (jdhuff.c process_restart): the bits left in the current byte are padding;
// the reader has necessarily run into the RSTn marker while filling (at most 7 data bits precede it), so step over it.  A
// stream that has something else there is corrupt: resynchronise on the next RSTn like libjpeg's default resync does.
JHD void jpeg_bits_restart(JpegBits* b) {
  jpeg_bits_fill(b);
  int64_t p = b->pos;
  if (b->marker >= 0xD0 && b->marker <= 0xD7) {
    p += 2;
  } else {
    while (p + 1 < b->end && !(b->d[p] == 0xFF && b->d[p + 1] >= 0xD0 && b->d[p + 1] <= 0xD7)) ++p;
    p = p + 1 < b->end ? p + 2 : b->end;
  }
  b->pos = p; b->buf = 0; b->cnt = 0; b->marker = 0;
}

// `nat`: the zigzag -> natural-order table (64 + 16 entries, jpeg_fill_natural_order) in whatever memory is cheap to index
// per lane — LDS on the GPU (a per-symbol lookup in global memory stalls the in-order wave for a full memory round trip).
template <typename NAT>
JHD void jpeg_fill_natural_order(NAT nat) {
  for (int k = 0; k < 80; ++k) nat[k] = (uint8_t)jpeg_natural_order(k);
}

// one 8x8 block: DC difference + AC run/size pairs -> coefficients in NATURAL order (the block must be zero on entry)
template <typename TDC, typename TAC, typename NAT>
JHD void jpeg_decode_block(JpegBits* b, TDC dc, TAC ac, NAT nat, int* dc_pred, int16_t* coef) {
  jpeg_bits_fill(b);
  int s = jpeg_decode_symbol(b, dc) & 15;
  if (s) *dc_pred = (int)((uint32_t)*dc_pred + (uint32_t)jpeg_receive_extend(b, s));   // (wraps, defined, on a hostile stream)
  coef[0] = (int16_t)*dc_pred;
  for (int k = 1; k < 64; ++k) {
    jpeg_bits_fill(b);
    const int rs = jpeg_decode_symbol(b, ac);
    const int r = rs >> 4;
    s = rs & 15;
    if (s) {
      k += r;
      const int v = jpeg_receive_extend(b, s);
      coef[nat[k]] = (int16_t)v;                      // k <= 63 + 15: entries past 63 alias 63 like libjpeg's table
    } else {
      if (r != 15) break;
      k += 15;
    }
  }
}

// ---------------------------------------------------------------------------------------------- progressive (jdphuff.c)
#ifndef JPEG_PROG_FAIL
#define JPEG_PROG_FAIL(code) return (code)       // (the host debug build redefines this to say where a file was given up)
#endif
// One bit / n bits of the entropy-coded segment.
JHD int jpeg_get_bits(JpegBits* b, int n) {            // n in 1..16
  jpeg_bits_fill(b);
  const int r = jpeg_bits_peek(b, n);
  jpeg_bits_skip(b, n);
  return r;
}
JHD int16_t jpeg_shl(int v, int al) { return (int16_t)(int32_t)((uint32_t)v << al); }

// refinement correction of an already-nonzero coefficient (decode_mcu_AC_refine): one bit; if set and the bit position is
// still clear, move the magnitude away from zero by 1 << Al
JHD void jpeg_refine_nonzero(JpegBits* b, int16_t* c, int p1, int m1) {
  if (jpeg_get_bits(b, 1)) {
    const int v = *c;
    if ((v & p1) == 0) *c = (int16_t)(v >= 0 ? v + p1 : v + m1);
  }
}

// Walks the marker segments from the first SOS on and decodes every scan into the (zeroed) coefficient planes c0 / c1 / c2
// (natural order, row stride = mcus_x * hs[c] blocks, like the sequential decoder's).  `tab.dc(id)` / `tab.ac(id)`, id 0 | 1, are
// the table views (rebuilt per scan from the latest DHT definitions: progressive files redefine tables between scans).
// Returns 0 when the file ended at EOI with coefficients 0-9 of every component fully refined; 1 for anything this decoder
// does not follow (table ids > 1, tables missing, quantisation tables redefined between scans, malformed scan headers, data
// that does not end at a marker); 2 when the scans stop early (libjpeg smooths such files: the caller lets it).
template <typename TAB, typename NAT>
JHD int jpeg_decode_progressive(const uint8_t* d, int64_t len, const JpegInfo* o, TAB tab, NAT nat, int16_t* c0, int16_t* c1,
                                int16_t* c2) {
  int64_t p = o->scan_off;
  int rst = o->restart_interval;
  int32_t dh_dc0 = o->dht_off[0], dh_dc1 = o->dht_off[1], dh_ac0 = o->dht_off[4], dh_ac1 = o->dht_off[5];
  uint64_t done0 = 0, done1 = 0, done2 = 0;            // bit k: coefficient k (zigzag) of the component has reached Al = 0
  int nscans = 0;
  const int ncomp = o->ncomp, W = o->width, H = o->height, hmax = o->hmax, vmax = o->vmax, mcus_x = o->mcus_x, mcus_y = o->mcus_y;
  for (int guard = 0; guard < 4096; ++guard) {
    if (p + 2 > len || d[p] != 0xFF) JPEG_PROG_FAIL(1);
    while (p < len && d[p] == 0xFF) ++p;
    if (p >= len) JPEG_PROG_FAIL(1);
    const int m = d[p++];
    if (m == 0xD9) {                                    // EOI
      const uint64_t need = 0x3FFull;
      if ((done0 & need) != need) JPEG_PROG_FAIL(2);
      if (ncomp == 3 && ((done1 & need) != need || (done2 & need) != need)) JPEG_PROG_FAIL(2);
      return 0;
    }
    if ((m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (p + 2 > len) JPEG_PROG_FAIL(1);
    const int L = jpeg_u16(d + p);
    if (L < 2 || p + L > len) JPEG_PROG_FAIL(1);
    const uint8_t* s = d + p + 2;
    const int n = L - 2;
    if (m == 0xC4) {
      int q = 0;
      while (q < n) {
        if (q + 17 > n) JPEG_PROG_FAIL(1);
        const int tc = s[q] >> 4, id = s[q] & 15;
        int cnt = 0;
        for (int i = 0; i < 16; ++i) cnt += s[q + 1 + i];
        if (tc > 1 || id > 1 || cnt > 256 || (tc == 0 && cnt > 16) || q + 17 + cnt > n) JPEG_PROG_FAIL(1);
        const int32_t off = (int32_t)(p + 2 + q + 1);
        if (tc == 0) { if (id == 0) dh_dc0 = off; else dh_dc1 = off; }
        else { if (id == 0) dh_ac0 = off; else dh_ac1 = off; }
        q += 17 + cnt;
      }
    } else if (m == 0xDD) {
      if (n >= 2) rst = jpeg_u16(s);
    } else if (m == 0xDB || (m >= 0xC0 && m <= 0xCF && m != 0xC4)) {
      JPEG_PROG_FAIL(1);                                // new quantisation tables / a second frame header between scans
    } else if (m == 0xDA) {
      if (++nscans > 128) JPEG_PROG_FAIL(1);             // (an encoder writes ~10; a hostile file could keep a lane busy for minutes)
      const int ns = n >= 1 ? s[0] : 0;
      if (ns < 1 || ns > ncomp || n < 1 + 2 * ns + 3) JPEG_PROG_FAIL(1);
      // scan components as indices into the frame's component list (in frame order, as the standard requires)
      int sc0 = -1, sc1 = -1, sc2 = -1, td0 = 0, td1 = 0, td2 = 0, ta0 = 0;
      for (int j = 0; j < ns; ++j) {
        const int id = s[1 + 2 * j], tt = s[2 + 2 * j];
        int ci = -1;
        if (id == o->cid[0]) ci = 0; else if (ncomp == 3 && id == o->cid[1]) ci = 1; else if (ncomp == 3 && id == o->cid[2]) ci = 2;
        if (ci < 0 || (tt >> 4) > 1 || (tt & 15) > 1) JPEG_PROG_FAIL(1);
        if (j == 0) { sc0 = ci; td0 = tt >> 4; ta0 = tt & 15; }
        else if (j == 1) { if (ci <= sc0) JPEG_PROG_FAIL(1); sc1 = ci; td1 = tt >> 4; }
        else { if (ci <= sc1) JPEG_PROG_FAIL(1); sc2 = ci; td2 = tt >> 4; }
      }
      const uint8_t* t = s + 1 + 2 * ns;
      const int Ss = t[0], Se = t[1], Ah = t[2] >> 4, Al = t[2] & 15;
      if (Ss > Se || Se > 63 || Al > 13 || Ah > 13 || (Ss == 0 && Se != 0) || (Ss != 0 && ns != 1)) JPEG_PROG_FAIL(1);
      // tables this scan decodes with
      if (Ss == 0) {
        if (Ah == 0) {
          const bool use0 = td0 == 0 || (ns > 1 && td1 == 0) || (ns > 2 && td2 == 0);
          const bool use1 = td0 == 1 || (ns > 1 && td1 == 1) || (ns > 2 && td2 == 1);
          if ((use0 && dh_dc0 < 0) || (use1 && dh_dc1 < 0)) JPEG_PROG_FAIL(1);
          if (use0) jpeg_build_huff(d + dh_dc0, tab.dc(0));
          if (use1) jpeg_build_huff(d + dh_dc1, tab.dc(1));
        }
      } else {
        const int32_t off = ta0 == 0 ? dh_ac0 : dh_ac1;
        if (off < 0) JPEG_PROG_FAIL(1);
        jpeg_build_huff(d + off, tab.ac(ta0));
      }
      // refinement bookkeeping: which coefficients are now final
      {
        const uint64_t band = (Se == 63 ? ~0ull : ((1ull << (Se + 1)) - 1ull)) & ~((1ull << Ss) - 1ull);
        for (int j = 0; j < ns; ++j) {
          const int ci = j == 0 ? sc0 : (j == 1 ? sc1 : sc2);
          uint64_t& dn = ci == 0 ? done0 : (ci == 1 ? done1 : done2);
          dn = Al == 0 ? (dn | band) : (dn & ~band);
        }
      }
      JpegBits b;
      jpeg_bits_init(&b, d, p + L, len);
      int pred0 = 0, pred1 = 0, pred2 = 0, togo = rst;
      unsigned eobrun = 0;
      const int p1 = 1 << Al, m1 = -(1 << Al);
      if (ns > 1) {
        // ---- interleaved scan (DC only): whole MCUs, dummy blocks included
        for (int my = 0; my < mcus_y; ++my)
          for (int mx = 0; mx < mcus_x; ++mx) {
            if (rst && togo == 0) { jpeg_bits_restart(&b); pred0 = pred1 = pred2 = 0; togo = rst; }
            for (int j = 0; j < ns; ++j) {
              const int ci = j == 0 ? sc0 : (j == 1 ? sc1 : sc2);
              const int tdj = j == 0 ? td0 : (j == 1 ? td1 : td2);
              const int hsj = ci == 0 ? o->hs[0] : (ci == 1 ? o->hs[1] : o->hs[2]);
              const int vsj = ci == 0 ? o->vs[0] : (ci == 1 ? o->vs[1] : o->vs[2]);
              int16_t* base = ci == 0 ? c0 : (ci == 1 ? c1 : c2);
              int& pred = ci == 0 ? pred0 : (ci == 1 ? pred1 : pred2);
              const int bw = mcus_x * hsj;
              for (int v = 0; v < vsj; ++v)
                for (int h = 0; h < hsj; ++h) {
                  int16_t* blk = base + ((long long)(my * vsj + v) * bw + mx * hsj + h) * 64;
                  if (Ah == 0) {
                    jpeg_bits_fill(&b);
                    const int sz = jpeg_decode_symbol(&b, tab.dc(tdj)) & 15;
                    if (sz) pred = (int)((uint32_t)pred + (uint32_t)jpeg_receive_extend(&b, sz));
                    blk[0] = jpeg_shl(pred, Al);
                  } else if (jpeg_get_bits(&b, 1)) {
                    blk[0] = (int16_t)(blk[0] | p1);
                  }
                }
            }
            if (rst) --togo;
          }
      } else {
        // ---- one component: its own block grid (no dummy blocks), one block per "MCU"
        const int ci = sc0;
        const int hsj = ci == 0 ? o->hs[0] : (ci == 1 ? o->hs[1] : o->hs[2]);
        const int vsj = ci == 0 ? o->vs[0] : (ci == 1 ? o->vs[1] : o->vs[2]);
        int16_t* base = ci == 0 ? c0 : (ci == 1 ? c1 : c2);
        const int bw = mcus_x * hsj;
        const int wb = (int)(((long long)W * hsj + (long long)hmax * 8 - 1) / ((long long)hmax * 8));
        const int hb = (int)(((long long)H * vsj + (long long)vmax * 8 - 1) / ((long long)vmax * 8));
        int pred = 0;
        for (int by = 0; by < hb; ++by)
          for (int bx = 0; bx < wb; ++bx) {
            if (rst && togo == 0) { jpeg_bits_restart(&b); pred = 0; eobrun = 0; togo = rst; }
            int16_t* blk = base + ((long long)by * bw + bx) * 64;
            if (Ss == 0) {
              if (Ah == 0) {
                jpeg_bits_fill(&b);
                const int sz = jpeg_decode_symbol(&b, tab.dc(td0)) & 15;
                if (sz) pred = (int)((uint32_t)pred + (uint32_t)jpeg_receive_extend(&b, sz));
                blk[0] = jpeg_shl(pred, Al);
              } else if (jpeg_get_bits(&b, 1)) {
                blk[0] = (int16_t)(blk[0] | p1);
              }
            } else if (Ah == 0) {
              // decode_mcu_AC_first
              if (eobrun > 0) {
                --eobrun;
              } else {
                for (int k = Ss; k <= Se; ++k) {
                  jpeg_bits_fill(&b);
                  const int rs = jpeg_decode_symbol(&b, tab.ac(ta0));
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz) {
                    k += r;
                    const int v = jpeg_receive_extend(&b, sz);
                    blk[nat[k]] = jpeg_shl(v, Al);            // k <= 63 + 15: entries past 63 alias 63 like libjpeg's table
                  } else if (r == 15) {
                    k += 15;
                  } else {
                    eobrun = 1u << r;
                    if (r) eobrun += (unsigned)jpeg_get_bits(&b, r);
                    --eobrun;
                    break;
                  }
                }
              }
            } else {
              // decode_mcu_AC_refine
              int k = Ss;
              if (eobrun == 0) {
                for (; k <= Se; ++k) {
                  jpeg_bits_fill(&b);
                  const int rs = jpeg_decode_symbol(&b, tab.ac(ta0));
                  int r = rs >> 4, sv = rs & 15;
                  if (sv) {
                    sv = jpeg_get_bits(&b, 1) ? p1 : m1;        // (size must be 1: libjpeg warns and goes on alike)
                  } else if (r != 15) {
                    eobrun = 1u << r;
                    if (r) eobrun += (unsigned)jpeg_get_bits(&b, r);
                    break;                                    // force end-of-band
                  }
                  // advance over already-nonzero coefficients and r still-zero ones, appending correction bits to the nonzeroes
                  do {
                    int16_t* c = blk + nat[k];
                    if (*c != 0) jpeg_refine_nonzero(&b, c, p1, m1);
                    else if (--r < 0) break;                  // reached the target zero coefficient
                    ++k;
                  } while (k <= Se);
                  if (sv) blk[nat[k]] = (int16_t)sv;
                }
              }
              if (eobrun > 0) {
                // correction bits for the already-nonzero coefficients after the end of band
                for (; k <= Se; ++k) {
                  int16_t* c = blk + nat[k];
                  if (*c != 0) jpeg_refine_nonzero(&b, c, p1, m1);
                }
                --eobrun;
              }
            }
            if (rst) --togo;
          }
      }
      // the scan's data must stop at a marker (< 8 padding bits before it)
      jpeg_bits_fill(&b);
      if (!b.marker || b.pos + 2 > len) JPEG_PROG_FAIL(1);
      p = b.pos;                                          // at the 0xFF of the next marker
      continue;
    }
    p += L;
  }
  JPEG_PROG_FAIL(1);
}

// ---------------------------------------------------------------------------------------------- IDCT (jidctint.c, ISLOW)
// in: 64 coefficients (natural order), q: 64 quantisation values (natural order); out: 8 rows of 8 samples, row stride `ld`.
// Final clamp = saturation to [0, 255] after the +128 level shift: what libjpeg-turbo's SIMD kernels compute (the C code's
// masked table lookup differs only for |value| > 511, which no encoder produces).
// All sums and products are taken modulo 2^32 (unsigned arithmetic, reinterpreted as two's complement where a sign matters):
// what libjpeg-turbo's 32-bit SIMD lanes compute, identical to its C code whenever nothing overflows — i.e. for every
// stream an encoder can produce — and DEFINED behaviour for the garbage coefficients of a damaged file.
typedef uint32_t ju32;
JHD int jpeg_descale(ju32 x, int n) { return (int)(int32_t)(x + ((ju32)1 << (n - 1))) >> n; }
JHD uint8_t jpeg_clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

JHD void jpeg_idct_1d(const int in[8], int out[8], int shift, bool first) {
  const ju32 C0298 = 2446, C0390 = 3196, C0541 = 4433, C0765 = 6270, C0899 = 7373, C1175 = 9633, C1501 = 12299, C1847 = 15137,
             C1961 = 16069, C2053 = 16819, C2562 = 20995, C3072 = 25172;
  (void)first;
  const ju32 i0 = (ju32)in[0], i1 = (ju32)in[1], i2 = (ju32)in[2], i3 = (ju32)in[3], i4 = (ju32)in[4], i5 = (ju32)in[5],
             i6 = (ju32)in[6], i7 = (ju32)in[7];
  ju32 z2 = i2, z3 = i6;
  ju32 z1 = (z2 + z3) * C0541;
  ju32 tmp2 = z1 - z3 * C1847;
  ju32 tmp3 = z1 + z2 * C0765;
  ju32 tmp0 = (i0 + i4) << 13;                 // << CONST_BITS
  ju32 tmp1 = (i0 - i4) << 13;
  const ju32 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = i7; tmp1 = i5; tmp2 = i3; tmp3 = i1;
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  ju32 z4 = tmp1 + tmp3;
  const ju32 z5 = (z3 + z4) * C1175;
  tmp0 *= C0298; tmp1 *= C2053; tmp2 *= C3072; tmp3 *= C1501;
  z1 = (ju32)0 - z1 * C0899; z2 = (ju32)0 - z2 * C2562; z3 = (ju32)0 - z3 * C1961; z4 = (ju32)0 - z4 * C0390;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  out[0] = jpeg_descale(tmp10 + tmp3, shift); out[7] = jpeg_descale(tmp10 - tmp3, shift);
  out[1] = jpeg_descale(tmp11 + tmp2, shift); out[6] = jpeg_descale(tmp11 - tmp2, shift);
  out[2] = jpeg_descale(tmp12 + tmp1, shift); out[5] = jpeg_descale(tmp12 - tmp1, shift);
  out[3] = jpeg_descale(tmp13 + tmp0, shift); out[4] = jpeg_descale(tmp13 - tmp0, shift);
}

JHD void jpeg_idct_block(const int16_t* coef, const uint16_t* q, uint8_t* out, int ld) {
  int ws[64];
#pragma unroll
  for (int c = 0; c < 8; ++c) {               // pass 1: columns, scaled up by 2^PASS1_BITS (2)
    int in[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = (int)((ju32)(int32_t)coef[r * 8 + c] * (ju32)q[r * 8 + c]);
    jpeg_idct_1d(in, o, 13 - 2, true);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r * 8 + c] = o[r];
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {               // pass 2: rows, descale by 2^(13 + 2 + 3), level shift, clamp
    int in[8], o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) in[c] = ws[r * 8 + c];
    jpeg_idct_1d(in, o, 13 + 2 + 3, false);
#pragma unroll
    for (int c = 0; c < 8; ++c) out[r * ld + c] = jpeg_clamp255((int)((ju32)o[c] + 128u));
  }
}

// ---------------------------------------------------------------------------------------------- upsample + colour
// chroma sample for output pixel (x, y) of a plane `pl` (row stride ld) holding dw x dh REAL samples, subsampled by
// (hsub, vsub) in {(1,1), (2,1), (2,2)}: jdsample.c fullsize / h2v1_fancy / h2v2_fancy, with libjpeg's edge rules (the row
// above the first = the first, below the last real row = the last; first / last column special cases).
JHD int jpeg_upsampled(const uint8_t* pl, int ld, int dw, int dh, int hsub, int vsub, int x, int y) {
  if (hsub == 1 && vsub == 1) return pl[y * ld + x];
  if (vsub == 1) {                                     // h2v1
    const int i = x >> 1;
    const uint8_t* r = pl + y * ld;
    if ((x & 1) == 0) return i == 0 ? r[0] : (3 * r[i] + r[i - 1] + 1) >> 2;
    return i == dw - 1 ? r[i] : (3 * r[i] + r[i + 1] + 2) >> 2;
  }
  const int j = y >> 1;
  const int jn = (y & 1) == 0 ? (j > 0 ? j - 1 : 0) : (j + 1 < dh ? j + 1 : dh - 1);      // the further row
  const uint8_t* r0 = pl + j * ld;
  const uint8_t* r1 = pl + jn * ld;
  const int i = x >> 1;                                // h2v2
  const int t = 3 * r0[i] + r1[i];
  if ((x & 1) == 0) {
    if (i == 0) return (t * 4 + 8) >> 4;
    return (t * 3 + (3 * r0[i - 1] + r1[i - 1]) + 8) >> 4;
  }
  if (i == dw - 1) return (t * 4 + 7) >> 4;
  return (t * 3 + (3 * r0[i + 1] + r1[i + 1]) + 7) >> 4;
}

JHD void jpeg_ycc_to_rgb(int y, int cb, int cr, uint8_t* rgb) {
  cb -= 128; cr -= 128;
  rgb[0] = jpeg_clamp255(y + ((91881 * cr + 32768) >> 16));
  rgb[1] = jpeg_clamp255(y + (((-22554) * cb + 32768 + (-46802) * cr) >> 16));
  rgb[2] = jpeg_clamp255(y + ((116130 * cb + 32768) >> 16));
}
