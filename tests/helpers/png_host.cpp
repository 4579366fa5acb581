// Host build of csrc/png_core.h for tests/test_png_core_host.py: a SERIAL writer made of the same functions the gfx950 kernels
// (csrc/png.hip) run in parallel — filter choice, code lengths, canonical codes, block header, bit packing, Adler-32, CRC-32 by
// pieces + combine — so the format logic is checked against PIL's reader on the build host.  Test infrastructure only.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../domain-rag_amd/csrc/png_core.h"

static uint32_t crc_bytes(const uint8_t* p, size_t n) {
  uint32_t c = 0xffffffffu;
  for (size_t i = 0; i < n; ++i) c = png_crc_table_entry((c ^ p[i]) & 0xff) ^ (c >> 8);
  return c ^ 0xffffffffu;
}

// img: [H, W, C] uint8 (C = 1 or 3).  piece: bytes per CRC piece (the pieces are combined with png_crc_combine).
// Returns the file size, or -1 when cap is too small.
extern "C" int64_t png_host_encode(const uint8_t* img, int H, int W, int C, uint8_t* out, int64_t cap, int piece) {
  const int64_t rb = (int64_t)W * C, nf = (int64_t)H * (1 + rb);
  std::vector<uint8_t> filt((size_t)nf);
  std::vector<uint32_t> count(PNG_NSYM, 0);
  for (int r = 0; r < H; ++r) {
    const uint8_t* cur = img + (int64_t)r * rb;
    const uint8_t* prev = r ? cur - rb : nullptr;
    int64_t cost[5] = {0, 0, 0, 0, 0};
    for (int f = 0; f < 5; ++f)
      for (int64_t x = 0; x < rb; ++x) {
        const int a = x >= C ? cur[x - C] : 0, b = prev ? prev[x] : 0, c = (prev && x >= C) ? prev[x - C] : 0;
        cost[f] += png_filter_cost(png_filter_byte(f, cur[x], a, b, c));
      }
    int best = 0;
    for (int f = 1; f < 5; ++f) if (cost[f] < cost[best]) best = f;
    uint8_t* o = filt.data() + (int64_t)r * (1 + rb);
    o[0] = (uint8_t)best;
    for (int64_t x = 0; x < rb; ++x) {
      const int a = x >= C ? cur[x - C] : 0, b = prev ? prev[x] : 0, c = (prev && x >= C) ? prev[x - C] : 0;
      o[1 + x] = png_filter_byte(best, cur[x], a, b, c);
    }
  }
  for (int64_t i = 0; i < nf; ++i) count[filt[i]]++;
  count[256] = 1;
  uint8_t len[PNG_NSYM];
  std::vector<int32_t> work(4 * 520);
  png_code_lengths(count.data(), len, work.data());
  uint32_t code[PNG_NSYM];
  png_canonical_codes(len, code);
  uint64_t bits = PNG_HEADER_BITS;
  for (int s = 0; s < PNG_NSYM; ++s) bits += (uint64_t)count[s] * len[s];
  const int64_t D = (int64_t)((bits + 7) / 8), Z = 2 + D + 4, total = PNG_FILE_PREFIX + Z + 4 + 12;
  if (total > cap) return -1;
  std::vector<uint32_t> words((size_t)(D / 4 + 3), 0);
  png_block_header(len, words.data());
  uint32_t pos = PNG_HEADER_BITS;
  for (int64_t i = 0; i <= nf; ++i) {
    const uint32_t c = code[i < nf ? filt[i] : 256];
    png_put_bits(words.data(), pos, c & 0xffff, (int)(c >> 16));
  }
  // Adler-32 of the filtered stream in the closed form the kernels use: A = 1 + sum d, B = n + sum (n - j) d_j  (mod 65521)
  uint64_t sa = 0, sb = 0;
  for (int64_t j = 0; j < nf; ++j) { sa += filt[j]; sb = (sb + (uint64_t)(nf - j) % 65521 * filt[j]) % 65521; }
  const uint32_t A = (uint32_t)((1 + sa) % 65521), B = (uint32_t)((nf % 65521 + sb) % 65521);
  const uint32_t adler = (B << 16) | A;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  memcpy(out, sig, 8);
  png_ihdr_chunk(W, H, C, out + 8);
  uint8_t* p = out + 33;
  p[0] = (uint8_t)(Z >> 24); p[1] = (uint8_t)(Z >> 16); p[2] = (uint8_t)(Z >> 8); p[3] = (uint8_t)Z;
  memcpy(p + 4, "IDAT", 4);
  p[8] = 0x78; p[9] = 0x01;
  memcpy(p + 10, words.data(), (size_t)D);            // little-endian words == LSB-first byte stream
  uint8_t* q = p + 10 + D;
  q[0] = (uint8_t)(adler >> 24); q[1] = (uint8_t)(adler >> 16); q[2] = (uint8_t)(adler >> 8); q[3] = (uint8_t)adler;
  // CRC over type + data, by pieces
  const uint8_t* cs = p + 4;
  const int64_t cn = 4 + Z;
  uint32_t crc = 0;
  for (int64_t o = 0; o < cn; o += piece) {
    const int64_t m = cn - o < piece ? cn - o : piece;
    const uint32_t c2 = crc_bytes(cs + o, (size_t)m);
    crc = o == 0 ? c2 : png_crc_combine(crc, c2, (uint64_t)m);
  }
  q[4] = (uint8_t)(crc >> 24); q[5] = (uint8_t)(crc >> 16); q[6] = (uint8_t)(crc >> 8); q[7] = (uint8_t)crc;
  static const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xae, 0x42, 0x60, 0x82};
  memcpy(q + 8, iend, 12);
  return total;
}

// code lengths alone, for the length-limit test (counts -> lengths)
extern "C" void png_host_code_lengths(const uint32_t* count, uint8_t* len) {
  std::vector<int32_t> work(4 * 520);
  png_code_lengths(count, len, work.data());
}
