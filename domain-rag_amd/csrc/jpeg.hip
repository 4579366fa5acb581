// jpeg.hip — baseline JPEG files -> RGB pixels on gfx950, byte-identical to PIL / libjpeg-turbo (arithmetic: jpeg_core.h).
//
// Replaces the `Image.open(path).convert("RGB")` the reference pays per corpus image before CLIP's preprocess
// (retrieval/clip100_resnet_style_all_shots.py:270-281) — SURVEY §8(f)-2: once the tower embeds 11 k images/s, libjpeg on the
// host cores (3.8 k images/s with 32 worker processes) is what bounds stage 1.
//
// The host only reads the files: a batch is one byte blob + offsets.  Four launches per batch:
//   jpeg_parse_kernel      one image per LANE: walks the marker segments, writes a 48-word descriptor (size, sampling, table
//                          offsets, scan offset, status).  The host reads the descriptors back once to size the outputs.
//   jpeg_huffman_kernel    one image per LANE, one wave per workgroup.  Entropy decoding is inherently serial per image (COCO
//                          files carry no restart markers), so the parallelism is ACROSS images: 64 bit-streams per wave, every
//                          lane with its own four code tables in LDS (direct table + packed length limits + symbols, 85 KiB per
//                          wave, lane-interleaved so a lookup is bank-conflict free whatever the codes are; no per-symbol access
//                          leaves LDS / registers), coefficients scattered into a zeroed int16
//                          buffer in natural order.  A 4096-image batch is 64 waves on 64 CUs; the other CUs keep running the
//                          embedding tower of the previous batch on another stream.
//   jpeg_idct_kernel       one thread per 8x8 block: dequantise + ISLOW IDCT in registers, 8 x 8-byte row stores into the
//                          component plane.
//   jpeg_color_kernel      one thread per output pixel: fancy chroma upsampling + YCbCr -> RGB, 3-byte store.
// The last two are plain data-parallel integer kernels bound by the load/store units (no reuse to stage in LDS).
#include "drag_common.h"
#include "jpeg_core.h"

namespace {

// Per-lane decoding tables in LDS, element e of lane l at base[e * 64 + l] (the 64 lanes of an access hit 64 consecutive
// elements: no bank conflicts whatever the codes are).  6-bit direct tables; 16 symbols per DC table, 256 per AC table.
template <int LB_, int NV_>
struct LdsTable {
  enum { LB = LB_, NV = NV_ };
  DRAG_LDS uint16_t* l;
  DRAG_LDS uint32_t* k;
  DRAG_LDS uint8_t* v;
  __device__ __forceinline__ DRAG_LDS uint16_t& lut(int i) const { return l[i * 64]; }
  __device__ __forceinline__ DRAG_LDS uint32_t& limk(int i) const { return k[i * 64]; }
  __device__ __forceinline__ DRAG_LDS uint8_t& val(int i) const { return v[i * 64]; }
};
typedef LdsTable<6, 16> DcTable;
typedef LdsTable<6, 256> AcTable;      // 6-bit direct tables keep the wave at 85 KiB of LDS: see JPEG_HUFF_LDS

struct JpegArgs {
  const uint8_t* data;
  const int64_t* off;       // [n + 1] byte offsets of the files inside `data`
  const JpegInfo* info;     // [n]
  const int64_t* plan;      // [n, 3]: coefficient offset (int16 elements), plane offset (bytes), output offset (bytes)
  int16_t* coef;
  uint8_t* planes;
  uint16_t* qtab;           // [n, 3, 64] quantisation tables in natural order
  uint8_t* out;
  int32_t* scan_status;     // [n]: 0 = the entropy-coded data ended at the EOI marker, as a clean file's does
  int n;
};

__global__ __launch_bounds__(64) void jpeg_parse_kernel(const uint8_t* data, const int64_t* off, int n, JpegInfo* info) {
  // the descriptor is indexed dynamically while it is built (component / table numbers come from the file), so a per-lane local
  // copy lives in scratch memory (196 B / lane in rounds 1-2); one LDS slot per lane instead: 12 KiB per wave, no scratch
  __shared__ JpegInfo slot[64];
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  JpegInfo& o = slot[threadIdx.x];
  jpeg_parse(data + off[i], off[i + 1] - off[i], &o);
  if (o.status == 0) {                               // the Huffman kernel keeps four code tables per lane: DC 0/1, AC 0/1,
    const uint8_t* d = data + off[i];                // and 16 symbol slots per DC table (8-bit JPEG has at most 12 categories)
    if (o.progressive) {
      // td / ta belong to the scans (jpeg_parse stops at the first SOS and leaves them 0): what can be checked here is every DC table
      // defined BEFORE the first scan — jpeg_decode_progressive builds them unchecked (it checks the ones defined between scans itself)
      for (int id = 0; id < 2; ++id) {
        if (o.dht_off[id] < 0) continue;
        int cnt = 0;
        for (int l = 0; l < 16; ++l) cnt += d[o.dht_off[id] + l];
        if (cnt > 16) { o.status = JPEG_ERR_TABLES; break; }
      }
    } else {
      for (int c = 0; c < o.ncomp; ++c) {
        if (o.td[c] > 1 || o.ta[c] > 1) { o.status = JPEG_ERR_TABLES; break; }
        int cnt = 0;
        for (int l = 0; l < 16; ++l) cnt += d[o.dht_off[o.td[c]] + l];
        if (cnt > 16) { o.status = JPEG_ERR_TABLES; break; }
      }
    }
  }
  info[i] = o;
}

struct LdsNat {                      // zigzag -> natural order, one copy per wave
  DRAG_LDS uint8_t* t;
  __device__ __forceinline__ DRAG_LDS uint8_t& operator[](int k) const { return t[k]; }
};

// LDS carve (bytes): u16 direct tables [4 x 64][64 lanes] | u32 limit/first words [4 x 17][64] |
//                    u8 symbols [2 x 16 DC + 2 x 256 AC][64] | natural order [80]
// The direct tables are deliberately short (6 bits): with 64 streams per wave some lane is on a longer code at almost every
// symbol, so the long-code path (16 packed words, fetched unconditionally) runs anyway and a bigger table buys nothing — while
// 85 KiB instead of 131 lets the wave share a CU's 160 KiB with the embedding tower's workgroups, so that the decode of one
// chunk overlaps the tower of the previous one instead of waiting for a CU to drain completely.
constexpr int JPEG_LUT_ELEMS = 4 * 64;
constexpr int JPEG_LUT_BYTES = JPEG_LUT_ELEMS * 64 * 2;        //  32 768
constexpr int JPEG_LIMK_BYTES = 4 * 17 * 64 * 4;               //  17 408
constexpr int JPEG_VAL_BYTES = (2 * 16 + 2 * 256) * 64;        //  34 816
constexpr int JPEG_HUFF_LDS = JPEG_LUT_BYTES + JPEG_LIMK_BYTES + JPEG_VAL_BYTES + 128;   // 85 120

__global__ __launch_bounds__(64) void jpeg_huffman_kernel(JpegArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x;
  const int i = blockIdx.x * 64 + lane;
  const LdsNat nat{(DRAG_LDS uint8_t*)lds + JPEG_LUT_BYTES + JPEG_LIMK_BYTES + JPEG_VAL_BYTES};
  for (int k = lane; k < 80; k += 64) nat[k] = (uint8_t)jpeg_natural_order(k);
  __syncthreads();
  if (i >= a.n) return;
  const JpegInfo& o = a.info[i];
  if (o.status != 0) { a.scan_status[i] = 0; return; }
  if (o.progressive) return;                         // jpeg_progressive_kernel's file
  const uint8_t* d = a.data + a.off[i];
  const int64_t len = a.off[i + 1] - a.off[i];
  DRAG_LDS uint16_t* const L = (DRAG_LDS uint16_t*)lds + lane;
  DRAG_LDS uint32_t* const K = (DRAG_LDS uint32_t*)(lds + JPEG_LUT_BYTES) + lane;
  DRAG_LDS uint8_t* const V = (DRAG_LDS uint8_t*)(lds + JPEG_LUT_BYTES + JPEG_LIMK_BYTES) + lane;
  // table id 0 / 1 of each class
  const DcTable dct[2] = {{L, K, V}, {L + 64 * 64, K + 17 * 64, V + 16 * 64}};
  const AcTable act[2] = {{L + 128 * 64, K + 34 * 64, V + 32 * 64}, {L + (128 + 64) * 64, K + 51 * 64, V + (32 + 256) * 64}};
#pragma unroll
  for (int id = 0; id < 2; ++id) {
    if (o.dht_off[id] >= 0) jpeg_build_huff(d + o.dht_off[id], dct[id]);
    if (o.dht_off[4 + id] >= 0) jpeg_build_huff(d + o.dht_off[4 + id], act[id]);
  }
  // per-component constants with STATIC indices (a dynamically indexed local array lives in scratch = global memory, and a
  // scratch access per block costs this one-wave-per-CU kernel a full memory round trip)
  const int ncomp = o.ncomp, mcus_x = o.mcus_x, mcus_y = o.mcus_y, rst = o.restart_interval;
  int hs[3], vs[3], bw[3], td[3], ta[3];
  int16_t* cbase[3];
  {
    long long p = a.plan[(long long)i * 3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const bool on = c < ncomp;
      hs[c] = on ? o.hs[c] : 0; vs[c] = on ? o.vs[c] : 0;
      td[c] = on ? (o.td[c] & 1) : 0; ta[c] = on ? (o.ta[c] & 1) : 0;
      bw[c] = mcus_x * hs[c];
      cbase[c] = a.coef + p;
      p += (long long)bw[c] * (mcus_y * vs[c]) * 64;
      if (on) {                                      // quantisation table, natural order, for the IDCT kernel
        const uint8_t* qt = d + o.dqt_off[o.tq[c]];
        const bool q16 = o.dqt_16[o.tq[c]] != 0;
        uint16_t* q = a.qtab + ((long long)i * 3 + c) * 64;
        for (int k = 0; k < 64; ++k) q[nat[k]] = q16 ? (uint16_t)jpeg_u16(qt + 2 * k) : (uint16_t)qt[k];
      }
    }
  }
  JpegBits b;
  jpeg_bits_init(&b, d, o.scan_off, len);
  int pred[3] = {0, 0, 0};
  int togo = rst;
  for (int my = 0; my < mcus_y; ++my)
    for (int mx = 0; mx < mcus_x; ++mx) {
      if (rst && togo == 0) {
        jpeg_bits_restart(&b);
        pred[0] = pred[1] = pred[2] = 0;
        togo = rst;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // the view of table id 1 is the view of id 0 plus constant offsets: select by arithmetic, not by indexing an array
        const DcTable dc{L + td[c] * (64 * 64), K + td[c] * (17 * 64), V + td[c] * (16 * 64)};
        const AcTable ac{L + (128 + ta[c] * 64) * 64, K + (34 + ta[c] * 17) * 64, V + (32 + ta[c] * 256) * 64};
        for (int v = 0; v < vs[c]; ++v)
          for (int h = 0; h < hs[c]; ++h) {
            int16_t* blk = cbase[c] + ((long long)(my * vs[c] + v) * bw[c] + mx * hs[c] + h) * 64;
            jpeg_decode_block(&b, dc, ac, nat, &pred[c], blk);
          }
      }
      if (rst) --togo;
    }
  // A clean scan ends with < 8 padding bits followed by EOI.  Anything else (data cut short, trailing segments, a decoder
  // that lost sync on damaged data) is reported: libjpeg / PIL decide what such a file means (warning, OSError), not this kernel.
  jpeg_bits_fill(&b);
  a.scan_status[i] = (b.marker == 0xD9 && b.pos + 2 <= len) ? 0 : 1;
}

// SOF2 files (round 3): one file per lane like the sequential kernel, same LDS table geometry; the scans are walked by
// jpeg_decode_progressive (jpeg_core.h), which rebuilds the lane's tables whenever a scan needs them.  Launched over the whole
// batch: lanes whose file is not progressive leave at once (a batch without progressive files pays one empty launch).
struct LaneTables {
  DRAG_LDS uint16_t* L;
  DRAG_LDS uint32_t* K;
  DRAG_LDS uint8_t* V;
  __device__ __forceinline__ DcTable dc(int id) const { return DcTable{L + id * (64 * 64), K + id * (17 * 64), V + id * (16 * 64)}; }
  __device__ __forceinline__ AcTable ac(int id) const { return AcTable{L + (128 + id * 64) * 64, K + (34 + id * 17) * 64, V + (32 + id * 256) * 64}; }
};

__global__ __launch_bounds__(64) void jpeg_progressive_kernel(JpegArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x;
  const int i = blockIdx.x * 64 + lane;
  const LdsNat nat{(DRAG_LDS uint8_t*)lds + JPEG_LUT_BYTES + JPEG_LIMK_BYTES + JPEG_VAL_BYTES};
  for (int k = lane; k < 80; k += 64) nat[k] = (uint8_t)jpeg_natural_order(k);
  __syncthreads();
  if (i >= a.n) return;
  const JpegInfo& o = a.info[i];
  if (o.status != 0 || !o.progressive) return;
  const uint8_t* d = a.data + a.off[i];
  const int64_t len = a.off[i + 1] - a.off[i];
  const LaneTables tab{(DRAG_LDS uint16_t*)lds + lane, (DRAG_LDS uint32_t*)(lds + JPEG_LUT_BYTES) + lane,
                       (DRAG_LDS uint8_t*)(lds + JPEG_LUT_BYTES + JPEG_LIMK_BYTES) + lane};
  int16_t* cbase[3];
  {
    long long p = a.plan[(long long)i * 3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const bool on = c < o.ncomp;
      cbase[c] = a.coef + p;
      if (on) {
        p += (long long)(o.mcus_x * o.hs[c]) * (o.mcus_y * o.vs[c]) * 64;
        const uint8_t* qt = d + o.dqt_off[o.tq[c]];
        const bool q16 = o.dqt_16[o.tq[c]] != 0;
        uint16_t* q = a.qtab + ((long long)i * 3 + c) * 64;
        for (int k = 0; k < 64; ++k) q[nat[k]] = q16 ? (uint16_t)jpeg_u16(qt + 2 * k) : (uint16_t)qt[k];
      }
    }
  }
  const int rc = jpeg_decode_progressive(d, len, &o, tab, nat, cbase[0], cbase[1], cbase[2]);
  a.scan_status[i] = rc;                              // 0 clean; 1 not followed / damaged; 2 scans stop early (libjpeg would smooth)
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(JpegArgs a) {
  const int i = blockIdx.y;
  const JpegInfo& o = a.info[i];
  if (o.status != 0) return;
  long long blk = (long long)blockIdx.x * 256 + threadIdx.x;
  long long cofs = a.plan[(long long)i * 3], pofs = a.plan[(long long)i * 3 + 1];
  int c = 0;
  for (; c < o.ncomp; ++c) {
    const long long nb = (long long)(o.mcus_x * o.hs[c]) * (o.mcus_y * o.vs[c]);
    if (blk < nb) break;
    blk -= nb; cofs += nb * 64; pofs += nb * 64;
  }
  if (c == o.ncomp) return;
  const int bwc = o.mcus_x * o.hs[c];
  const int by = (int)(blk / bwc), bx = (int)(blk - (long long)by * bwc);
  const int ld = bwc * 8;
  jpeg_idct_block(a.coef + cofs + blk * 64, a.qtab + ((long long)i * 3 + c) * 64, a.planes + pofs + (long long)by * 8 * ld + bx * 8, ld);
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(JpegArgs a) {
  const int i = blockIdx.y;
  const JpegInfo& o = a.info[i];
  if (o.status != 0) return;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  const int W = o.width, H = o.height;
  if (px >= (long long)W * H) return;
  const int y = (int)(px / W), x = (int)(px - (long long)y * W);
  const uint8_t* p0 = a.planes + a.plan[(long long)i * 3 + 1];
  uint8_t* dst = a.out + a.plan[(long long)i * 3 + 2] + px * 3;
  const int ld0 = o.mcus_x * o.hs[0] * 8;
  const int Y = p0[(long long)y * ld0 + x];
  if (o.ncomp == 1) { dst[0] = dst[1] = dst[2] = (uint8_t)Y; return; }
  const long long n0 = (long long)ld0 * (o.mcus_y * o.vs[0] * 8);
  const int ld1 = o.mcus_x * 8;
  const long long n1 = (long long)ld1 * (o.mcus_y * 8);
  const int dw = (W + o.hmax - 1) / o.hmax, dh = (H + o.vmax - 1) / o.vmax;
  const int cb = jpeg_upsampled(p0 + n0, ld1, dw, dh, o.hmax, o.vmax, x, y);
  const int cr = jpeg_upsampled(p0 + n0 + n1, ld1, dw, dh, o.hmax, o.vmax, x, y);
  uint8_t rgb[3];
  jpeg_ycc_to_rgb(Y, cb, cr, rgb);
  dst[0] = rgb[0]; dst[1] = rgb[1]; dst[2] = rgb[2];
}

}  // namespace

static_assert(sizeof(JpegInfo) == sizeof(drag_jpeg_info), "drag_jpeg_info must mirror JpegInfo");

extern "C" int drag_jpeg_parse(const void* data, const int64_t* offsets, int32_t n, drag_jpeg_info* info, void* stream) {
  DRAG_CHECK(data && offsets && info, "drag_jpeg_parse: null pointer");
  DRAG_CHECK(n > 0, "drag_jpeg_parse: n must be positive");
  hipLaunchKernelGGL(jpeg_parse_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const uint8_t*)data, offsets, n,
                     (JpegInfo*)info);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_jpeg_decode_rgb(const void* data, const int64_t* offsets, const drag_jpeg_info* info, const int64_t* plan,
                                    int32_t n, int64_t max_blocks, int64_t max_pixels, void* coef_ws, int64_t coef_bytes,
                                    void* plane_ws, void* qtab_ws, void* out_rgb, int32_t* scan_status, void* stream) {
  DRAG_CHECK(data && offsets && info && plan && coef_ws && plane_ws && qtab_ws && out_rgb && scan_status,
             "drag_jpeg_decode_rgb: null pointer");
  DRAG_CHECK(n > 0 && max_blocks > 0 && max_pixels > 0 && coef_bytes > 0, "drag_jpeg_decode_rgb: bad sizes");
  DRAG_CHECK(max_blocks < (1ll << 31) * 256 && max_pixels < (1ll << 31) * 256 && n <= 65535, "drag_jpeg_decode_rgb: batch too large");
  hipStream_t st = (hipStream_t)stream;
  JpegArgs a;
  a.data = (const uint8_t*)data; a.off = offsets; a.info = (const JpegInfo*)info; a.plan = plan;
  a.coef = (int16_t*)coef_ws; a.planes = (uint8_t*)plane_ws; a.qtab = (uint16_t*)qtab_ws; a.out = (uint8_t*)out_rgb; a.scan_status = scan_status; a.n = n;
  hipError_t e = hipMemsetAsync(coef_ws, 0, (size_t)coef_bytes, st);     // blocks are sparse: only non-zero coefficients are stored
  DRAG_CHECK(e == hipSuccess, "drag_jpeg_decode_rgb: memset failed");
  const int lds = JPEG_HUFF_LDS;                                          // 85 KiB per wave
  static bool lds_ok = false;
  if (!lds_ok) {
    e = hipFuncSetAttribute((const void*)jpeg_huffman_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DRAG_CHECK(e == hipSuccess, "drag_jpeg_decode_rgb: cannot raise the dynamic LDS limit to 128 KiB");
    lds_ok = true;
  }
  hipLaunchKernelGGL(jpeg_huffman_kernel, dim3((n + 63) / 64), dim3(64), lds, st, a);
  DRAG_LAUNCH_CHECK();
  static bool lds_ok2 = false;
  if (!lds_ok2) {
    e = hipFuncSetAttribute((const void*)jpeg_progressive_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DRAG_CHECK(e == hipSuccess, "drag_jpeg_decode_rgb: cannot raise the dynamic LDS limit (progressive kernel)");
    lds_ok2 = true;
  }
  hipLaunchKernelGGL(jpeg_progressive_kernel, dim3((n + 63) / 64), dim3(64), lds, st, a);     // SOF2 files; other lanes leave at once
  DRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((max_blocks + 255) / 256), n), dim3(256), 0, st, a);
  DRAG_LAUNCH_CHECK();
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((max_pixels + 255) / 256), n), dim3(256), 0, st, a);
  DRAG_LAUNCH_CHECK();
  return 0;
}
