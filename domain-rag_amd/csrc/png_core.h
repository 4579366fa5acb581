// png_core.h — the arithmetic of a PNG writer whose files decode to exactly the pixels handed in: row filters (PNG 1.2 §6),
// a Huffman-only DEFLATE stream (RFC 1951 dynamic block, literals + end-of-block, no LZ77 matches), Adler-32 (RFC 1950) and
// CRC-32 with the GF(2) combine that lets chunks be checksummed in parallel.  Replaces the reference's `image.save(path)` of
// its large RGB results (batch_generate_flux_kshot.py:480 generated_image_rank{r}.png; outpainting_updown_sampling_redux.py:
// 1262 *_hires_result_*.png, :1278 *_final_result_*.png): PIL writes zlib level 6 there; a PNG is defined by the pixels it
// decodes to, not by its bytes, so the contract is "PIL (any reader) decodes the same array".  Photographic content is
// entropy-coded, not dictionary-coded, by zlib too (few matches survive the filters): Huffman-only lands within ~10 % of level 6
// on generated images; flat synthetic images (masks) are better served by the host's zlib and stay there.
//
// Host/device neutral like jpeg_core.h: csrc/png.hip runs these functions in gfx950 kernels, tests/helpers/png_host.cpp
// compiles the same header with g++ and writes whole files serially, which the CPU tests decode with PIL.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PHD __host__ __device__ __forceinline__
#else
#define PHD inline
#endif

enum {
  PNG_NSYM = 257,                 // literals 0..255 + end-of-block (256); no length codes are ever used
  PNG_MAXBITS = 15,
  // block header: BFINAL(1) BTYPE(2) HLIT(5) HDIST(5) HCLEN(4) + 19 x 3 code-length-code lengths + 258 lengths x 4 bits
  PNG_HEADER_BITS = 3 + 5 + 5 + 4 + 19 * 3 + 258 * 4,
  PNG_FILE_PREFIX = 8 + 25 + 8,   // signature + IHDR chunk + IDAT length/type: the zlib stream starts here
};

// ---- filters (bytes of one row; a = left, b = up, c = up-left, all 0 outside the image) ----
PHD int png_paeth(int a, int b, int c) {
  const int p = a + b - c;
  const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
PHD uint8_t png_filter_byte(int f, int x, int a, int b, int c) {
  switch (f) {
    case 0: return (uint8_t)x;
    case 1: return (uint8_t)(x - a);
    case 2: return (uint8_t)(x - b);
    case 3: return (uint8_t)(x - ((a + b) >> 1));
    default: return (uint8_t)(x - png_paeth(a, b, c));
  }
}
// libpng's selection heuristic: the filter whose output, read as signed bytes, has the smallest sum of magnitudes
PHD int png_filter_cost(uint8_t v) { return v < 128 ? v : 256 - v; }

// ---- Huffman code lengths for the 257 symbols ------------------------------------------------------------------------
// Two-queue construction over the symbols sorted by (count, symbol); when the tree is deeper than 15 the counts are halved
// (floor 1) and the tree rebuilt — deterministic, and within a fraction of a percent of the optimal length-limited code.
// The sort is a separate step so that a kernel can do it with all its lanes (rank sort: same order, by definition).
// Scratch: freq[514], par[514], sym[257] int32.
PHD int png_sort_symbols(const int32_t* cnt, int32_t* freq, int32_t* sym) {       // serial form; returns the number of used symbols
  int n = 0;
  for (int s = 0; s < PNG_NSYM; ++s)
    if (cnt[s] > 0) {                         // insertion sort by (count, symbol): symbols arrive in increasing order, ties keep it
      int i = n++;
      while (i > 0 && freq[i - 1] > cnt[s]) { freq[i] = freq[i - 1]; sym[i] = sym[i - 1]; --i; }
      freq[i] = cnt[s]; sym[i] = s;
    }
  return n;
}
// leaves freq[0..n) sorted ascending with their symbols -> len[sym] = depth; returns the deepest leaf (len is written only
// when that is <= PNG_MAXBITS)
PHD int png_tree_lengths(int n, int32_t* freq, int32_t* par, const int32_t* sym, uint8_t* len) {
  if (n == 1) { len[sym[0]] = 1; return 1; }  // (cannot happen for an image: the end-of-block symbol always joins a literal)
  int leaf = 0, inner = n, made = n;          // queue heads; nodes [n, made) are internal, created in non-decreasing weight
  while (made < 2 * n - 1) {
    int pick[2];
    for (int k = 0; k < 2; ++k) {
      const bool take_leaf = leaf < n && (inner >= made || freq[leaf] <= freq[inner]);
      pick[k] = take_leaf ? leaf++ : inner++;
    }
    freq[made] = freq[pick[0]] + freq[pick[1]];
    par[pick[0]] = made; par[pick[1]] = made;
    ++made;
  }
  // depths: a node's parent has a larger index, so walking down from the root every parent is finished first;
  // depth[i] overwrites par[i]
  par[made - 1] = 0;
  int maxd = 0;
  for (int i = made - 2; i >= 0; --i) {
    par[i] = par[par[i]] + 1;
    if (i < n && par[i] > maxd) maxd = par[i];
  }
  if (maxd <= PNG_MAXBITS)
    for (int i = 0; i < n; ++i) len[sym[i]] = (uint8_t)par[i];
  return maxd;
}
PHD int32_t png_clamp_count(uint32_t c) { return (int32_t)(c > 0x3fffffu ? 0x3fffffu : c); }   // node weights (sums of up to 257 counts) stay below 2^31
// work: >= 4 * 520 int32
PHD void png_code_lengths(const uint32_t* count, uint8_t* len, int32_t* work) {
  int32_t* freq = work; int32_t* par = work + 520; int32_t* sym = work + 1040; int32_t* cnt = work + 1560;
  for (int s = 0; s < PNG_NSYM; ++s) { cnt[s] = png_clamp_count(count[s]); len[s] = 0; }
  for (;;) {
    const int n = png_sort_symbols(cnt, freq, sym);
    if (png_tree_lengths(n, freq, par, sym, len) <= PNG_MAXBITS) return;
    for (int s = 0; s < PNG_NSYM; ++s)
      if (cnt[s] > 0) cnt[s] = (cnt[s] + 1) >> 1;
  }
}

PHD uint32_t png_reverse_bits(uint32_t v, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
  return r;
}
// canonical codes (RFC 1951 §3.2.2), stored bit-reversed so that they can be OR-ed into an LSB-first bit stream:
// code[s] = (length << 16) | reversed code
PHD void png_canonical_codes(const uint8_t* len, uint32_t* code) {
  int bl_count[PNG_MAXBITS + 1];
  uint32_t next[PNG_MAXBITS + 2];
  for (int b = 0; b <= PNG_MAXBITS; ++b) bl_count[b] = 0;
  for (int s = 0; s < PNG_NSYM; ++s) bl_count[len[s]]++;
  bl_count[0] = 0;
  uint32_t c = 0;
  for (int b = 1; b <= PNG_MAXBITS; ++b) { c = (c + bl_count[b - 1]) << 1; next[b] = c; }
  for (int s = 0; s < PNG_NSYM; ++s) {
    const int l = len[s];
    code[s] = l ? ((uint32_t)l << 16) | png_reverse_bits(next[l]++, l) : 0u;
  }
}

// The dynamic block's header as PNG_HEADER_BITS bits, LSB first, into words[0 .. 34] (OR-ed: the caller zeroes them).
// Code-length alphabet: symbols 0..15 get 4-bit codes (a complete code: canonical code of symbol v is v), 16/17/18 are unused,
// so every literal length is written as 4 bits and the header has a FIXED size — the data's bit offsets do not depend on it.
PHD void png_put_bits(uint32_t* words, uint32_t& pos, uint32_t v, int n) {
  const uint32_t w = pos >> 5, sh = pos & 31;
  words[w] |= v << sh;
  if (sh + n > 32) words[w + 1] |= v >> (32 - sh);
  pos += n;
}
PHD void png_block_header(const uint8_t* len, uint32_t* words) {
  uint32_t pos = 0;
  png_put_bits(words, pos, 1, 1);             // BFINAL
  png_put_bits(words, pos, 2, 2);             // BTYPE = 10 (dynamic Huffman)
  png_put_bits(words, pos, 0, 5);             // HLIT: 257 literal/length codes
  png_put_bits(words, pos, 0, 5);             // HDIST: 1 distance code (of zero bits: no distance codes are used)
  png_put_bits(words, pos, 15, 4);            // HCLEN: all 19 code-length-code lengths
  const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  for (int i = 0; i < 19; ++i) png_put_bits(words, pos, order[i] < 16 ? 4 : 0, 3);
  for (int s = 0; s < PNG_NSYM + 1; ++s) {    // 257 literal lengths, then the one distance length (0)
    const uint32_t v = s < PNG_NSYM ? len[s] : 0;
    png_put_bits(words, pos, png_reverse_bits(v, 4), 4);
  }
}

// ---- checksums ----------------------------------------------------------------------------------------------------
// CRC-32 (ISO 3309, reflected, polynomial 0xEDB88320) — bytewise update and the combine of two adjacent pieces:
// crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B)   (the init / final inversions cancel in this form, as in zlib's crc32_combine).
PHD uint32_t png_crc_table_entry(uint32_t n) {
  uint32_t c = n;
  for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
  return c;
}
PHD uint32_t png_gf2_mulmod(uint32_t a, uint32_t b) {      // polynomials in reflected form: bit 31 = x^0
  uint32_t p = 0;
  for (uint32_t m = 0x80000000u; m; m >>= 1) {
    if (a & m) p ^= b;
    b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
  }
  return p;
}
PHD uint32_t png_x_pow_8n(uint64_t nbytes) {               // x^(8 n) mod P by square and multiply
  uint32_t r = 0x80000000u;                                 // x^0
  uint32_t sq = 0x00800000u;                                // x^8
  while (nbytes) {
    if (nbytes & 1) r = png_gf2_mulmod(sq, r);
    sq = png_gf2_mulmod(sq, sq);
    nbytes >>= 1;
  }
  return r;
}
PHD uint32_t png_crc_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
  return png_gf2_mulmod(png_x_pow_8n(len_b), crc_a) ^ crc_b;
}

// IHDR chunk (25 bytes: length, type, 13 data bytes, CRC) for an 8-bit grey (channels 1) or RGB (channels 3) image
inline void png_ihdr_chunk(int width, int height, int channels, uint8_t* out25) {
  const uint8_t d[25] = {0, 0, 0, 13, 'I', 'H', 'D', 'R',
                         (uint8_t)(width >> 24), (uint8_t)(width >> 16), (uint8_t)(width >> 8), (uint8_t)width,
                         (uint8_t)(height >> 24), (uint8_t)(height >> 16), (uint8_t)(height >> 8), (uint8_t)height,
                         8, (uint8_t)(channels == 1 ? 0 : 2), 0, 0, 0, 0, 0, 0, 0};
  uint32_t c = 0xffffffffu;
  for (int i = 4; i < 21; ++i) c = png_crc_table_entry((c ^ d[i]) & 0xff) ^ (c >> 8);
  c ^= 0xffffffffu;
  for (int i = 0; i < 21; ++i) out25[i] = d[i];
  out25[21] = (uint8_t)(c >> 24); out25[22] = (uint8_t)(c >> 16); out25[23] = (uint8_t)(c >> 8); out25[24] = (uint8_t)c;
}
