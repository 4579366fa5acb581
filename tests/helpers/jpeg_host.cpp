// Host build of csrc/jpeg_core.h for tests/test_jpeg_core_host.py: the SAME functions the gfx950 kernels call, compiled with
// g++ so their arithmetic can be compared with PIL's decode where there is no GPU.  Test infrastructure only.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../domain-rag_amd/csrc/jpeg_core.h"

extern "C" int jpeg_host_info(const uint8_t* d, int64_t len, int32_t* out48) {
  JpegInfo o;
  jpeg_parse(d, len, &o);
  memcpy(out48, &o, sizeof(o));
  return o.status;
}

// decode to RGB [H, W, 3]; returns the parse status (nothing is written unless it is 0)
extern "C" int jpeg_host_decode_rgb(const uint8_t* file, int64_t len, uint8_t* rgb) {
  // the reader's contract (jpeg_core.h): the blob starts on an 8-byte boundary and JPEG_TAIL_PAD readable bytes follow the file
  std::vector<uint64_t> blob((len + JPEG_TAIL_PAD + 7) / 8 + 1, 0);
  memcpy(blob.data(), file, (size_t)len);
  const uint8_t* d = (const uint8_t*)blob.data();
  JpegInfo o;
  jpeg_parse(d, len, &o);
  if (o.status) return o.status;
  // one table view per (component, class): plain arrays on the host
  struct HostTable {
    enum { LB = 6, NV = 256 };      // the kernel's table geometry (csrc/jpeg.hip)
    uint16_t* l; uint32_t* k; uint8_t* v;
    uint16_t& lut(int i) const { return l[i]; }
    uint32_t& limk(int i) const { return k[i]; }
    uint8_t& val(int i) const { return v[i]; }
  };
  struct HostTableDC : HostTable { enum { LB = 6, NV = 16 }; };      // the GPU kernel uses a 6-bit direct table for DC codes: same here
  std::vector<uint16_t> luts(6 * 256);
  std::vector<uint32_t> limk(6 * 17);
  std::vector<uint8_t> vals(6 * 256);
  // coefficient storage
  std::vector<std::vector<int16_t>> coef(3);
  for (int c = 0; c < o.ncomp; ++c) coef[c].assign((size_t)jpeg_blocks_w(&o, c) * jpeg_blocks_h(&o, c) * 64, 0);
  uint8_t nat[80];
  jpeg_fill_natural_order(nat);
  if (o.progressive) {
    // SOF2: the scan walker shared with jpeg_progressive_kernel; table id 0 / 1 of each class, rebuilt per scan
    struct HostTab {
      uint16_t* l; uint32_t* k; uint8_t* v;
      HostTableDC dc(int id) const { HostTableDC t; t.l = l + id * 256; t.k = k + id * 17; t.v = v + id * 256; return t; }
      HostTable ac(int id) const { HostTable t; t.l = l + (2 + id) * 256; t.k = k + (2 + id) * 17; t.v = v + (2 + id) * 256; return t; }
    };
    const HostTab tab{luts.data(), limk.data(), vals.data()};
    const int rc = jpeg_decode_progressive(d, len, &o, tab, (const uint8_t*)nat, coef[0].data(), o.ncomp == 3 ? coef[1].data() : nullptr,
                                           o.ncomp == 3 ? coef[2].data() : nullptr);
    if (rc) return 100 + rc;          // 101: not followed / damaged, 102: scans stop early (the caller lets libjpeg decide)
  } else {
  HostTableDC tdc[3];
  HostTable tac[3];
  for (int c = 0; c < o.ncomp; ++c) {
    tdc[c].l = luts.data() + (2 * c) * 256; tdc[c].k = limk.data() + (2 * c) * 17; tdc[c].v = vals.data() + (2 * c) * 256;
    tac[c].l = luts.data() + (2 * c + 1) * 256; tac[c].k = limk.data() + (2 * c + 1) * 17; tac[c].v = vals.data() + (2 * c + 1) * 256;
    jpeg_build_huff(d + o.dht_off[o.td[c]], tdc[c]);
    jpeg_build_huff(d + o.dht_off[4 + o.ta[c]], tac[c]);
  }
  JpegBits b;
  jpeg_bits_init(&b, d, o.scan_off, len);
  int pred[3] = {0, 0, 0};
  int togo = o.restart_interval;
  for (int my = 0; my < o.mcus_y; ++my)
    for (int mx = 0; mx < o.mcus_x; ++mx) {
      if (o.restart_interval && togo == 0) { jpeg_bits_restart(&b); pred[0] = pred[1] = pred[2] = 0; togo = o.restart_interval; }
      for (int c = 0; c < o.ncomp; ++c)
        for (int v = 0; v < o.vs[c]; ++v)
          for (int h = 0; h < o.hs[c]; ++h) {
            const int bx = mx * o.hs[c] + h, by = my * o.vs[c] + v;
            int16_t* blk = coef[c].data() + ((size_t)by * jpeg_blocks_w(&o, c) + bx) * 64;
            jpeg_decode_block(&b, tdc[c], tac[c], (const uint8_t*)nat, &pred[c], blk);
          }
      if (o.restart_interval) --togo;
    }
  }
  // planes
  std::vector<std::vector<uint8_t>> plane(o.ncomp);
  for (int c = 0; c < o.ncomp; ++c) {
    const int bw = jpeg_blocks_w(&o, c), bh = jpeg_blocks_h(&o, c), ld = bw * 8;
    plane[c].assign((size_t)ld * bh * 8, 0);
    uint16_t q[64];
    const uint8_t* qt = d + o.dqt_off[o.tq[c]];
    for (int k = 0; k < 64; ++k) q[jpeg_natural_order(k)] = o.dqt_16[o.tq[c]] ? (uint16_t)jpeg_u16(qt + 2 * k) : qt[k];
    for (int by = 0; by < bh; ++by)
      for (int bx = 0; bx < bw; ++bx)
        jpeg_idct_block(coef[c].data() + ((size_t)by * bw + bx) * 64, q, plane[c].data() + (size_t)by * 8 * ld + bx * 8, ld);
  }
  const int W = o.width, H = o.height;
  if (o.ncomp == 1) {
    const int ld = jpeg_blocks_w(&o, 0) * 8;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) { const uint8_t g = plane[0][(size_t)y * ld + x]; uint8_t* p = rgb + ((size_t)y * W + x) * 3; p[0] = p[1] = p[2] = g; }
    return 0;
  }
  const int ld0 = jpeg_blocks_w(&o, 0) * 8, ld1 = jpeg_blocks_w(&o, 1) * 8;
  const int dw = (W + o.hmax - 1) / o.hmax, dh = (H + o.vmax - 1) / o.vmax;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const int Y = plane[0][(size_t)y * ld0 + x];
      const int cb = jpeg_upsampled(plane[1].data(), ld1, dw, dh, o.hmax, o.vmax, x, y);
      const int cr = jpeg_upsampled(plane[2].data(), ld1, dw, dh, o.hmax, o.vmax, x, y);
      jpeg_ycc_to_rgb(Y, cb, cr, rgb + ((size_t)y * W + x) * 3);
    }
  return 0;
}

// decode with the output allocated from the PARSED size (mutated headers change it); returns a checksum, or -status.
// Used by the fuzz driver (tests/helpers/jpeg_fuzz.cpp, built with -fsanitize=address): whatever the bytes are, the shared
// arithmetic must stay inside its buffers — on the GPU an out-of-bounds access is a memory fault, not an exception.
extern "C" long long jpeg_host_decode_checked(const uint8_t* file, int64_t len, int64_t max_pixels) {
  int32_t info[48];
  const int st = jpeg_host_info(file, len, info);
  if (st) return -st;
  const int64_t px = (int64_t)info[1] * info[2];
  if (px > max_pixels) return -100;
  std::vector<uint8_t> rgb((size_t)px * 3);
  const int rc = jpeg_host_decode_rgb(file, len, rgb.data());
  if (rc) return -rc;
  long long sum = 0;
  for (size_t i = 0; i < rgb.size(); i += 97) sum += rgb[i];
  return sum;
}
