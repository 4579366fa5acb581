// attention.hip — Flux joint attention on gfx950: q/k RMSNorm + RoPE + V^T repack, then a
// flash-style MFMA attention (head_dim 128, bf16, no mask).
//
// Replaces FluxAttnProcessor2_0 (diffusers 0.33.1, un-vendored): norm_q/norm_k(/norm_added_*),
// apply_rotary_emb and F.scaled_dot_product_attention, reached on every denoise step from
// batch_generate_flux_kshot.py:467-474 and outpainting_updown_sampling_redux.py:1246-1257.
//
// Attention structure: block = 8 waves x 32 queries (4 waves for short sequences); KV tile = 64 keys; K tile [64][128] and
// V^T tile [128][64] arrive by LDS-DMA (buffer_load ... lds), double-buffered, one barrier per
// tile, XOR-swizzled on the source address + on the ds_read_b128.  QK^T is computed swapped
// (S^T = K Q^T, v_mfma_f32_32x32x16_bf16) so each lane owns ONE query column: the online
// softmax is lane-local apart from one lane<->lane^32 exchange, and the packed P registers are
// directly the B operand of O^T += V^T P^T.  The V^T image is stored with keys permuted inside
// groups of 16 so that a plain 16-byte read yields exactly the keys a lane's P registers hold
// (no transposes, no cross-lane traffic in the loop).
#include "drag_common.h"
// The asm statements that write m0 (one s_add_u32 m0 per LDS-DMA piece) list "m0" as a clobber: hipcc then re-materialises m0 before its own
// next LDS-DMA builtin (checked on a two-builtin probe: without the clobber the second builtin ran on the asm's stale m0).  clang warns that m0
// is a reserved register on every such statement; the clobber is what is wanted here.
#pragma clang diagnostic ignored "-Winline-asm"
#include "attn_q64_tile.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// --------------------------------------------------------------------------------------------
// q/k RMSNorm + RoPE in place, V -> V^T.   grid (ceil(S/64), H, B), block 256.
// --------------------------------------------------------------------------------------------
struct PrepArgs {
  bf16_t* qkv;
  bf16_t* vt;
  const bf16_t *wq_txt, *wk_txt, *wq_img, *wk_img;
  const float *cosT, *sinT;
  int B, S, H, ld, s_txt, s_pad;
  float eps;
  int skip_q;      // q is normalised / rotated by the attention kernel as it loads its Q fragments (drag_attention_qprep_bf16)
};

__global__ __launch_bounds__(256) void qk_norm_rope_vt_kernel(PrepArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t sv[64 * 136];  // V tile, row padded to 136 (272 B)
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int HD = p.H * 128;
  bf16_t* base = p.qkv + (long long)b * p.S * p.ld;

  // ---- q and k: 16 lanes per row, 8 elements per lane ----
  const int sub = tid & 15;        // which 8-element group of the 128
  const int rloc = tid >> 4;       // 0..15
  const bool do_norm = p.wq_txt != nullptr, do_rope = p.cosT != nullptr;
  if (do_norm || do_rope)
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    if (which == 0 && p.skip_q) continue;
    for (int it = 0; it < 4; ++it) {
      const int s = s0 + it * 16 + rloc;
      const bool ok = s < p.S;
      const int sc = ok ? s : p.S - 1;
      bf16_t* ptr = base + (long long)sc * p.ld + which * HD + h * 128 + sub * 8;
      const u32x4_t raw = *(const u32x4_t*)ptr;
      float x[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[2 * j] = bf2f((bf16_t)(raw[j] & 0xffff));
        x[2 * j + 1] = bf2f((bf16_t)(raw[j] >> 16));
      }
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
      ss += __shfl_xor(ss, 8, 64);
      ss += __shfl_xor(ss, 4, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 1, 64);
      const float rs = rsqrtf(ss * (1.0f / 128.0f) + p.eps);
      u32x4_t wr = (u32x4_t){0u, 0u, 0u, 0u};
      if (do_norm) {
        const bf16_t* wsel = which == 0 ? (sc < p.s_txt ? p.wq_txt : p.wq_img) : (sc < p.s_txt ? p.wk_txt : p.wk_img);
        wr = *(const u32x4_t*)(wsel + sub * 8);
      }
      f32x4_t c4 = (f32x4_t){1.f, 1.f, 1.f, 1.f}, s4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (do_rope) {
        c4 = *(const f32x4_t*)(p.cosT + (long long)sc * 64 + sub * 4);
        s4 = *(const f32x4_t*)(p.sinT + (long long)sc * 64 + sub * 4);
      }
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // diffusers RMSNorm: (x * rsqrt(var+eps)).to(bf16) * weight(bf16) -> bf16
        float a0 = x[2 * j], a1 = x[2 * j + 1];
        if (do_norm) {
          a0 = rbf(rbf(a0 * rs) * bf2f((bf16_t)(wr[j] & 0xffff)));
          a1 = rbf(rbf(a1 * rs) * bf2f((bf16_t)(wr[j] >> 16)));
        }
        // apply_rotary_emb (use_real, unbind_dim=-1): out = x*cos + rot(x)*sin in fp32
        const float r0 = a0 * c4[j] - a1 * s4[j];
        const float r1 = a1 * c4[j] + a0 * s4[j];
        o[j] = pack2bf(r0, r1);
      }
      if (ok) *(u32x4_t*)ptr = o;
    }
  }

  if (p.vt == nullptr) return;        // k only: the attention kernel reads V row-major
  // ---- v: stage [64 keys][128 d] in LDS, write V^T[d][pos(key)] ----
  for (int i = tid; i < 64 * 16; i += 256) {
    const int r = i >> 4, c = i & 15;
    const int s = s0 + r;
    u32x4_t v = (u32x4_t){0u, 0u, 0u, 0u};
    if (s < p.S) v = *(const u32x4_t*)(base + (long long)s * p.ld + 2 * HD + h * 128 + c * 8);
    *(u32x4_t*)(sv + r * 136 + c * 8) = v;
  }
  __syncthreads();
  bf16_t* vtb = p.vt + ((long long)(b * p.H + h) * 128) * p.s_pad + s0;
  for (int i = tid; i < 128 * 8; i += 256) {
    const int d = i >> 3, g8 = i & 7;  // 8 positions [8*g8, 8*g8+8) of row d
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      // position q = 8*g8 + j  ->  group = q>>4, within = q&15 = 8h+jj -> key offset 8*(jj>>2)+4h+(jj&3)
      const int q0 = 8 * g8 + j, q1 = q0 + 1;
      const int k0 = (q0 & ~15) + 8 * (((q0 & 15) & 7) >> 2) + 4 * ((q0 & 15) >> 3) + (q0 & 3);
      const int k1 = (q1 & ~15) + 8 * (((q1 & 15) & 7) >> 2) + 4 * ((q1 & 15) >> 3) + (q1 & 3);
      w[j >> 1] = (uint32_t)sv[k0 * 136 + d] | ((uint32_t)sv[k1 * 136 + d] << 16);
    }
    *(u32x4_t*)(vtb + (long long)d * p.s_pad + 8 * g8) = (u32x4_t){w[0], w[1], w[2], w[3]};
  }
}

// --------------------------------------------------------------------------------------------
// flash attention, D = 128.   grid (ceil(S/128), H, B), block 256.
// --------------------------------------------------------------------------------------------
struct AttnArgs {
  const bf16_t *q, *k, *vt;
  const bf16_t* v;            // VROW kernels: V as the Linear wrote it (rows of ld_qk elements, like k); vt is then unused
  bf16_t* out;
  int B, S, H, ld_qk, ld_o, s_pad;
  long long qk_bs, o_bs;
  float c;  // scale * log2(e)
  unsigned k_bytes, vt_bytes;          // span of ONE (batch, head) in the K tensor / the V^T tensor (attention_q64_kernel's descriptors)
  unsigned k_bytes_all, vt_bytes_all;  // span of the whole K tensor (all batches, all heads) / the whole V^T tensor (attention_d128_kernel)
  // optional Q preparation fused into the fragment load (all null / 0 = q is used as stored):
  const bf16_t *wq_txt, *wq_img;   // RMSNorm weights [128] of the text / image stream (rows < s_txt are text)
  const float *cosT, *sinT;        // RoPE tables [S, 64]
  int s_txt;
  float eps;
  int tune;       // measurement bits: 1 = static priority for the younger wave half, 2 = 16-byte epilogue stores
  int items;      // attention_q64_kernel: (batch-head, query block) items in all = the one-item grid; a smaller grid walks them (persistent)
};

constexpr float DEFER_THR = 8.0f;     // log2 units; 0 = rescale on every increase (classic online softmax)
constexpr int KT_BYTES = 64 * 256;   // K tile   [64 keys][128 d] bf16
constexpr int VT_BYTES = 128 * 128;  // V^T tile [128 d][64 keys] bf16

// SCHED 0: per KV tile, 16 {S(j+1) MFMA | exp of S(j)} steps, then the 16 P V MFMAs.
// SCHED 2 (experiment, "attn_sched" 3): SCHED 1 with every V^T fragment read one step ahead of its MFMA.
// SCHED 1: the P V MFMAs of key group s2 are issued as soon as that group's 16 probabilities are packed (from step 4 on, one
//          per step next to the S MFMA), so the exp stream is spread over 28 MFMAs instead of 16 and only 4 P V MFMAs trail
//          the loop.  Same arithmetic, same order per accumulator: bit-identical outputs.
// PMAX: the row maxima of S(j+1) are taken at the END of iteration j (next to the trailing P V MFMAs, which need no VALU)
//       instead of at the start of iteration j+1, where the matrix pipe has nothing to do.  Same values, same order.
// VROW: V is read ROW-MAJOR, straight from the qkv buffer (no V^T pass, no V^T buffer).  The V tile [64 keys][128 d] is staged like
//       the K tile (4 key rows per LDS-DMA instruction) and the A operand of O^T += V^T P^T — per lane 8 keys of one d — comes from
//       two ds_read_b64_tr_b16: within a 16-lane group source lane 4j + r supplies 4 consecutive d of key j, result lane 4r + e
//       receives (key 0..3) at d = 4r + e (probed on the hardware: scripts/probe/probe_tr16.hip).  The first read fetches the key
//       quad that this half-wave's P registers 0-3 hold, the second the quad of registers 4-7, so the key permutation the V^T
//       image needed is absorbed by the addresses.  16-byte units of a key row are XOR-swizzled with 4 * (key & 3): the 16 units
//       (4 keys x 64 B) a 32-lane access touches fall into 16 different bank groups.  Same MFMA operands -> same bits as the V^T kernels.
// PERSIST (round 3; measured 1 % slower than one item per workgroup, kept off the product path — see attention_launch): one workgroup per CU walks the items (batch-head, query block) loc = slot, slot + P/8, ... of its XCD's list and
//       the K / V^T LDS-DMA stream runs CONTINUOUSLY across them: the last two KV iterations of an item — whose prefetches used to
//       fetch tiles past the sequence end that nobody reads — fetch tiles 0, 0, 1 of the NEXT item instead, so the next item starts
//       with its operands in LDS and pays neither a workgroup dispatch nor the first HBM round trip.  Measured per workgroup outside
//       its KV loop before this (scripts/bench_attn_seams.py, B=8 S=5337): 6.6 us plain, 10.9 us with the q preparation = 4.3 % / 7.0 %
//       of an 84-tile workgroup.  The descriptors span the whole K / V^T tensors and an item is a scalar byte offset, so switching
//       items costs two SGPRs.  Same arithmetic per item: bit-identical outputs.
template <int NW, int SCHED, bool QPREP, bool PMAX, bool VROW = false, bool PERSIST = false>   // NW: waves per block (4 or 8), 32 queries each; QPREP: RMSNorm + RoPE of q on load
__global__ __launch_bounds__(512, 2) void attention_d128_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * KT_BYTES + 2 * VT_BYTES];
  const int w = wave_id(), l = lane_id();
  const int hh = l >> 5;            // half-wave
  // XCD-aware mapping (speed only): block id -> XCD id%8; all query blocks of one (batch, head) run on ONE XCD so
  // its K / V^T stream is fetched into a single private L2 instead of all eight.
  constexpr int QB = NW * 32;          // queries per block
  constexpr int CPW = 16 / NW;         // 1-KiB DMA chunks per wave per tile (K and V^T tiles have 16 each)
  const int nqb = (p.S + QB - 1) / QB;
  const int xcd = blockIdx.x & 7;
  int loc = blockIdx.x >> 3;           // index into this XCD's item list: item -> (bh = (loc / nqb) * 8 + xcd, query block loc % nqb)
  const int nloc = ((p.B * p.H + 7) / 8) * nqb;
  const int lstep = PERSIST ? (int)(gridDim.x >> 3) : nloc;
  if (((loc / nqb) * 8 + xcd) >= p.B * p.H) return;      // (PERSIST launches only when B * H is a multiple of 8: every item exists)
  int b, h, q0;
  unsigned soK, soV;                  // scalar byte offsets of this item's (batch, head) inside the K tensor / the V^T (or V) tensor
  auto item_offsets = [&](int lc, int& bb, int& hd, unsigned& ok, unsigned& ov) {
    const int bh = (lc / nqb) * 8 + xcd;
    bb = bh / p.H; hd = bh - bb * p.H;
    ok = (unsigned)(((long long)bb * p.qk_bs + hd * 128) * 2);
    ov = VROW ? ok : (unsigned)((((long long)(bb * p.H + hd)) * 128) * p.s_pad * 2);
  };
  item_offsets(loc, b, h, soK, soV);
  q0 = (loc - (loc / nqb) * nqb) * QB + w * 32;

  // ---- Q fragments stay in registers: B operand, lane -> query (l&31), k = 16ks + 8hh .. +8 (loaded below, after the
  // first K / V^T tiles have been requested: one workgroup per CU, so nothing else hides this prologue's latency) ----
  bf16x8_t qf[8];
  // ---- staging descriptors.  One item per workgroup (the product): based at the item's (batch, head), spanning that (batch, head) only —
  // tensors of any size work as long as ONE (batch, head) stays below 2 GiB (whole-tensor descriptors had put a 4 GiB cap on
  // B x S for the sake of the persistent experiment).  PERSIST: the WHOLE K / V^T tensor, the item is a scalar offset.
  __amdgpu_buffer_rsrc_t rsK, rsV;
  if constexpr (PERSIST) {
    rsK = __builtin_amdgcn_make_buffer_rsrc((void*)p.k, 0, p.k_bytes_all, 0x00020000);
    rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(VROW ? p.v : p.vt), 0, VROW ? p.k_bytes_all : p.vt_bytes_all, 0x00020000);
  } else {
    const long long kb = ((long long)b * p.qk_bs + h * 128) * 2;
    rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.k + kb), 0, p.k_bytes, 0x00020000);
    if constexpr (VROW) rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.v + kb), 0, p.k_bytes, 0x00020000);
    else rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.vt + (long long)(b * p.H + h) * 128 * p.s_pad * 2), 0, p.vt_bytes, 0x00020000);
    soK = 0; soV = 0;
  }
  unsigned soKn = soK, soVn = soV;    // the NEXT item's offsets (PERSIST; = this item's when there is none: harmless re-reads)
  // K chunk c (1 KiB) = key rows 4c..4c+3; lane: row 4c + (l>>4), physical slot l&15
  // V chunk c (1 KiB) = d rows 8c..8c+7;   lane: row 8c + (l>>3), physical slot l&7
  int krow[4];                        // (fixed-size: a template-dependent array bound here makes hipcc drop the host stub)
  unsigned kslot[4], voff[4];
  // VROW: V chunk c = key rows 4c..4c+3 like K; physical slot l&15 holds logical 16-byte unit (l&15) ^ 4*(key&3)
  const unsigned vrslot = (unsigned)(((l & 15) ^ (4 * ((l >> 4) & 3))) * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i >= CPW) break;
    const int c = w * CPW + i;
    krow[i] = c * 4 + (l >> 4);
    kslot[i] = (unsigned)(((l & 15) ^ (krow[i] & 15)) * 16);
    const int vrow = c * 8 + (l >> 3);
    const int vslot = (l & 7) ^ ((vrow >> 1) & 7);
    voff[i] = (unsigned)(((long long)vrow * p.s_pad + vslot * 8) * 2);
  }
  // tile index kv0 >= s_pad names tile kv0 - s_pad of the NEXT item (PERSIST): wave-uniform selects, no branches
  auto stage_k1 = [&](int buf, int kv0, int i) {
    const int c = w * CPW + i;
    const bool nx = PERSIST && kv0 >= p.s_pad;
    const int kvr = nx ? kv0 - p.s_pad : kv0;
    const int kr = min(kvr + krow[i], p.S - 1);
    const unsigned ko = (unsigned)((long long)kr * p.ld_qk * 2) + kslot[i];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (DRAG_LDS void*)((DRAG_LDS char*)smem + buf * KT_BYTES + c * 1024), 16, ko, nx ? soKn : soK, 0, 0);
  };
  auto stage_k = [&](int buf, int kv0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= CPW) break;
      stage_k1(buf, kv0, i);
    }
  };
  auto stage_v1 = [&](int buf, int kv0, int i) {
    const int c = w * CPW + i;
    const bool nx = PERSIST && kv0 >= p.s_pad;
    const int kvr = nx ? kv0 - p.s_pad : kv0;
    const unsigned so = nx ? soVn : soV;
    DRAG_LDS char* dV = (DRAG_LDS char*)smem + 2 * KT_BYTES + buf * VT_BYTES + c * 1024;
    if constexpr (VROW) {
      const int vr = min(kvr + krow[i], p.S - 1);          // rows past S are real rows: finite values under a zero probability
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (DRAG_LDS void*)dV, 16, (unsigned)((long long)vr * p.ld_qk * 2) + vrslot, so, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (DRAG_LDS void*)dV, 16, voff[i], so + (unsigned)(kvr * 2), 0, 0);
    }
  };
  auto stage_v = [&](int buf, int kv0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= CPW) break;
      stage_v1(buf, kv0, i);
    }
  };

  // ---- fragment read offsets ----
  // K: row (l&31)+32t, logical slot 2ks+hh, physical = logical ^ (row&15)
  const int krd = (l & 31) * 256;
  const int kx = l & 15;
  // V^T: row (l&31)+32dt, logical slot 2s+hh, physical = logical ^ ((row>>1)&7)
  const int vrd = (l & 31) * 128;
  const int vx = ((l & 31) >> 1) & 7;

  f32x16_t oacc[4];
  float m_run, l_run;

  const int nkv = p.s_pad / 64;
  f32x16_t scur[2], snext[2];
  {
  // ---- depth-1 pipeline: while the VALU works on the softmax of tile j, the matrix pipe already computes
  // S(j+1) = K(j+1) Q^T (independent of it), then O += P(j) V(j).  K therefore runs one tile ahead of V:
  // iteration j needs K(j+1) and V(j) in LDS and stages K(j+2), V(j+1).
  // The loop is unrolled by two with the buffer parity as a compile-time constant: every ds_read address is one of 12
  // loop-invariant VGPRs (8 K slots, 4 V^T slots: the XOR swizzle makes them lane-dependent) plus an immediate, the
  // S accumulators ping-pong between two register sets instead of being copied, and the first MFMA of a chain takes
  // C = 0 as an inline constant.
  // Measured alternatives (same box, interleaved runs, B=8 S=5337): a depth-2 pipeline that also overlaps P(j-1) V(j-1)
  // with the softmax of tile j (32 uniform {MFMA, exp} steps, fragment ring 3 deep) 1003 vs 1058 TFLOP/s for this loop;
  // this loop without its barrier +4 %, without barrier and vmcnt wait +4 % (synchronisation is not the limiter);
  // K / V^T staged through registers (buffer_load -> ds_write_b128 at the end of the iteration) instead of LDS-DMA
  // 1041 vs 1088 TFLOP/s; the two wave halves run half an iteration apart (softmax half of one SIMD partner beside the
  // P V half of the other, two barriers per tile, V^T triple-buffered) 1047 vs 1068.
  // PMC (B=8 S=5337): effective clock 1.70 GHz (power-limited; 2.4 nominal), MFMA pipe 52-59 % busy at that clock.
  int ak[8], av[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) ak[ks] = krd + (((2 * ks + hh) ^ kx) << 4);
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) av[s2] = vrd + (((2 * s2 + hh) ^ vx) << 4);
  // VROW: byte offset of this lane's 8-byte source piece for d block dt (32 rows of V^T): key 4 hh + (si>>2) of a quad pair,
  // 16-byte unit 4 dt + 2 g1 + ((si>>1)&1) swizzled with 4 * (key & 3), half (si & 1); + 4096 per 16-key group, + 2048 for the second quad
  int avd[4];
  {
    const int si = l & 15, g1 = (l >> 4) & 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      avd[dt] = (4 * hh + (si >> 2)) * 256 + ((4 * (dt ^ (si >> 2)) + 2 * g1 + ((si >> 1) & 1)) << 4) + (si & 1) * 8;
  }
  auto read_v = [&](const int s2, const int dt, const int vb) -> bf16x8_t {
    if constexpr (VROW) {
      typedef short s16x4_t __attribute__((ext_vector_type(4)));
      typedef short s16x8_t __attribute__((ext_vector_type(8)));
      DRAG_LDS char* a = (DRAG_LDS char*)smem + (avd[dt] + (vb + s2 * 4096));
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((DRAG_LDS s16x4_t*)a);
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((DRAG_LDS s16x4_t*)(a + 2048));
      return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    } else {
      return *(const bf16x8_t*)(smem + (av[s2] + (vb + dt * (32 * 128))));
    }
  };
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  stage_k(0, 0);
  stage_v(0, 0);
  if (nkv > 1) stage_k(1, 64);
  for (;;) {                           // items of this workgroup (one, unless PERSIST)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  m_run = -INFINITY; l_run = 0.f;
  const bool has_next = PERSIST && loc + lstep < nloc;
  if (PERSIST) {                       // where the stream goes when this item's tiles run out
    int bn, hn;
    if (has_next) item_offsets(loc + lstep, bn, hn, soKn, soVn);
    else { soKn = soK; soVn = soV; }
  }
  // Q load (+ RMSNorm / RoPE when QPREP) while those tiles are in flight
  {
    const int qr = min(q0 + (l & 31), p.S - 1);
    const bf16_t* qp = p.q + (long long)b * p.qk_bs + (long long)qr * p.ld_qk + h * 128 + hh * 8;
    if constexpr (!QPREP) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8_t*)(qp + ks * 16);
    } else {
      // per-head RMSNorm(128) + interleaved RoPE of this lane's query row, with the rounding points of the separate pass
      // (qk_norm_rope_vt_kernel): rbf(rbf(x * rs) * w), rotation in fp32, one rounding to bf16.  The lane holds 64 of the
      // row's 128 elements (d = 16 ks + 8 hh + 0..7), lane ^ 32 the other 64; RoPE pairs (2j, 2j+1) never straddle lanes.
      // All 32 loads are issued before anything is consumed: one memory latency, overlapped with the K / V^T prologue DMA
      // (a per-ks `if (table != null)` form compiled to eight serialised load-wait-branch rounds: +91 us per B=8 launch).
      const bf16_t* wsel = (qr < p.s_txt ? p.wq_txt : p.wq_img) + hh * 8;
      const float* cp = p.cosT + (long long)qr * 64 + hh * 4;
      const float* sp = p.sinT + (long long)qr * 64 + hh * 4;
      u32x4_t raw[8], wr[8];
      f32x4_t c4[8], s4[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        raw[ks] = *(const u32x4_t*)(qp + ks * 16);
        wr[ks] = *(const u32x4_t*)(wsel + ks * 16);
        c4[ks] = *(const f32x4_t*)(cp + ks * 8);
        s4[ks] = *(const f32x4_t*)(sp + ks * 8);
      }
      float ss = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a0 = bf2f((bf16_t)(raw[ks][j] & 0xffff)), a1 = bf2f((bf16_t)(raw[ks][j] >> 16));
          ss += a0 * a0;
          ss += a1 * a1;
        }
      ss += __shfl_xor(ss, 32, 64);
      const float rs = rsqrtf(ss * (1.0f / 128.0f) + p.eps);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a0 = rbf(rbf(bf2f((bf16_t)(raw[ks][j] & 0xffff)) * rs) * bf2f((bf16_t)(wr[ks][j] & 0xffff)));
          const float a1 = rbf(rbf(bf2f((bf16_t)(raw[ks][j] >> 16)) * rs) * bf2f((bf16_t)(wr[ks][j] >> 16)));
          o[j] = pack2bf(a0 * c4[ks][j] - a1 * s4[ks][j], a1 * c4[ks][j] + a0 * s4[ks][j]);
        }
        qf[ks] = __builtin_bit_cast(bf16x8_t, o);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8_t kf = *(const bf16x8_t*)(smem + (ak[ks] + t * (32 * 256)));
      scur[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? zero16 : scur[t], 0, 0, 0);
    }
  // one KV tile; PAR = it & 1 is a compile-time constant; reads S from sc, writes S(it+1) to sn
  // keys >= S of a ragged last tile do not exist: -inf before the row maximum is taken
  auto mask_and_max = [&](f32x16_t (&sx)[2], const int kvs) -> float {
    if (kvs + 64 > p.S) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kvs + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= p.S) sx[t][r] = -INFINITY;
        }
    }
    float m = sx[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sx[t][r]);
    return fmaxf(m, __shfl_xor(m, 32, 64));
  };
  float mt_carry = 0.f;
  if (PMAX) mt_carry = mask_and_max(scur, 0);
  if (p.tune & 1) {       // T5 static form: the second-dispatched wave half loses VALU arbitration to the older half on every segment
    if (w >= NW / 2) __builtin_amdgcn_s_setprio(1);
  }
  auto body = [&](const int it, auto par, f32x16_t (&sc)[2], f32x16_t (&sn)[2]) {
    constexpr int PAR = decltype(par)::value;
    constexpr int KN = ((PAR + 1) & 1) * KT_BYTES;                 // K(it+1)
    constexpr int VB = 2 * KT_BYTES + PAR * VT_BYTES;              // V(it)
    const int kv0 = it * 64;
    // K(it+1), V(it) landed (issued one iteration ago); every wave is done with K(it) and V(it-1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the LDS-DMA for K(it+2) / V(it+1) is issued piecewise between the MFMAs of the interleaved loop below:
    //  a burst of 2*CPW buffer_load..lds per wave right after the barrier idles the matrix pipe of every SIMD)
    // ---- online softmax (exp2 domain), deferred rescale ----
    const float mt = PMAX ? mt_carry : mask_and_max(sc, kv0);
    // Only move the running max (and rescale O, l) when some row's max grew by more than 2^8 in the
    // exp2 domain; otherwise P = exp2(s - m_old) is bounded by 2^8, which fp32 accumulation and the
    // bf16 P operand (relative precision is scale-free) absorb.  The previous tile's P·V is complete
    // at this point, so everything still at the old scale (O, l) is rescaled exactly once.
    if (!__all((mt - m_run) * p.c <= DEFER_THR)) {
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    // ---- matrix pipe: S(it+1) = K(it+1) Q^T   ||   VALU: P(it) = exp2(S(it) - m), row sums, bf16 packing.
    // The two streams are independent, but hipcc emits 16 back-to-back MFMAs followed by the whole softmax, so the
    // interleave is written out: 16 steps of {prefetch next K fragment, 1 MFMA, 2 exps + pack}, each fenced with
    // sched_barrier so the order survives.  S(it+1) is computed unconditionally (in the last iteration it reads a
    // stale K buffer and is discarded).
    const float nmc = -(m_run * p.c);
    float ps = 0.f;
    u32x4_t pk[4];
    {
      bf16x8_t kf = *(const bf16x8_t*)(smem + (ak[0] + KN));
      bf16x8_t vpipe;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const bf16x8_t kcur = kf;
        if (g < 15) {
          const int t1 = (g + 1) >> 3, ks1 = (g + 1) & 7;
          kf = *(const bf16x8_t*)(smem + (ak[ks1] + (KN + t1 * (32 * 256))));
        }
        bf16x8_t vfg;
        if (SCHED == 1 && g >= 4) vfg = read_v((g >> 2) - 1, g & 3, VB);
        if (SCHED == 2) {      // the V^T fragment of step g was read in step g-1: a whole step (2 MFMAs + the softmax slice) covers its latency
          vfg = vpipe;
          if (g >= 3 && g < 15) vpipe = read_v(((g + 1) >> 2) - 1, (g + 1) & 3, VB);
        }
        sn[g >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kcur, qf[g & 7], (g & 7) == 0 ? zero16 : sn[g >> 3], 0, 0, 0);
        // unconditional (no branches: a branch here splits the block and hipcc then hoists the whole softmax out of the
        // interleave): past the last tile the K rows clamp to S-1 and the V^T offsets fall outside the descriptor's range
        // (reads return 0); both land in buffers nobody reads any more
        if (g < CPW) stage_k1(PAR, kv0 + 128, g);
        else if (g < 2 * CPW) stage_v1((PAR + 1) & 1, kv0 + 64, g - CPW);
        // the softmax arithmetic is side-effect free, so instruction selection is free to cluster all of it ahead of the
        // first step (it did: the steps then held only ds_read + MFMA).  Two empty asm volatile statements chain its
        // inputs after the previous step and its results before this step's fence.
        float y0, y1;      // s * c - m*c; written as asm so that the chain anchor costs no register copies
        asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %5"
                     : "=&v"(y0), "=&v"(y1) : "v"(sc[g >> 3][(2 * g) & 15]), "v"(sc[g >> 3][((2 * g) & 15) + 1]), "s"(p.c), "v"(nmc));
        const float e0 = __builtin_amdgcn_exp2f(y0);
        const float e1 = __builtin_amdgcn_exp2f(y1);
        ps += e0 + e1;
        uint32_t wd = pack2bf(e0, e1);
        asm volatile("" : "+v"(wd), "+v"(ps));
        pk[g >> 2][g & 3] = wd;
        if (SCHED >= 1 && g >= 4)       // O^T[dt = g & 3] += V^T[.., keys of group (g >> 2) - 1] P^T: that group was packed >= 1 step ago
          oacc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfg, __builtin_bit_cast(bf16x8_t, pk[(g >> 2) - 1]), oacc[g & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    l_run += ps;
    bf16x8_t pf[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) pf[s2] = __builtin_bit_cast(bf16x8_t, pk[s2]);
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int s2 = (SCHED >= 1 ? 3 : 0); s2 < 4; ++s2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8_t vf = read_v(s2, dt, VB);
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], oacc[dt], 0, 0, 0);
      }
    // S(it+1) is complete: its row maxima now, beside the trailing P V MFMAs (in the last iteration sn is stale and unused)
    if (PMAX) mt_carry = mask_and_max(sn, kv0 + 64);
  };
  for (int it = 0; it < nkv; it += 2) {
    body(it, std::integral_constant<int, 0>{}, scur, snext);
    if (it + 1 < nkv) body(it + 1, std::integral_constant<int, 1>{}, snext, scur);
  }

  // ---- epilogue: lane holds O[query l&31][d = 32dt + 8(r>>2) + 4hh + (r&3)] ----
  const float lt = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / lt;
  const int qrow = q0 + (l & 31);
  if (p.tune & 2) {
    // T21: lanes l and l + 32 hold the two 8-byte halves of each 16-byte output chunk.  One v_permlane32_swap per dword
    // regroups a pair of chunks (g, g + 1) so that the lower half-wave stores chunk g and the upper one chunk g + 1 as
    // whole 16-byte pieces: 8 store instructions per lane instead of 16 (the tail is store-ISSUE bound).
    bf16_t* op = p.out + (long long)b * p.o_bs + (long long)qrow * p.ld_o + h * 128 + 8 * hh;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        uint32_t a0 = pack2bf(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv);
        uint32_t a1 = pack2bf(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
        uint32_t b0 = pack2bf(oacc[dt][4 * g + 4] * inv, oacc[dt][4 * g + 5] * inv);
        uint32_t b1 = pack2bf(oacc[dt][4 * g + 6] * inv, oacc[dt][4 * g + 7] * inv);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        // lower lanes: (r[0], r[1]) = (own chunk-g half, upper lane's chunk-g half); upper lanes: (lower's chunk g+1 half, own)
        const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
        if (qrow < p.S) *(u32x4_t*)(op + 32 * dt + 8 * g) = o;
      }
  } else if (qrow < p.S) {
    bf16_t* op = p.out + (long long)b * p.o_bs + (long long)qrow * p.ld_o + h * 128 + 4 * hh;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t o;
        o[0] = pack2bf(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv);
        o[1] = pack2bf(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
        *(u32x2_t*)(op + 32 * dt + 8 * g) = o;
      }
  }
  if (!has_next) break;
  // ---- next item: its K(0), V^T(0), K(1) tiles are already in LDS / in flight (issued by the last two KV iterations) ----
  loc += lstep;
  item_offsets(loc, b, h, soK, soV);
  q0 = (loc - (loc / nqb) * nqb) * QB + w * 32;
  }                                    // items
  }
}


// --------------------------------------------------------------------------------------------
// attention_q64_kernel — 4 waves x 64 queries, ONE wave per SIMD with the whole 512-entry register file: the kernel of the long
// sequences (S >= 4096: the DiT's joint attention) since round 4.
// Each wave owns two 32-query groups and runs both against every K / V^T fragment it reads, which halves the LDS fragment
// reads per flop.  Same arithmetic per query group, same deferred-rescale decisions (taken per group), so the outputs equal the 8-wave
// kernel's bit for bit (test_attention_schedules_are_bit_identical).  Why it wins (profiles/r04_pmc_attention_8wave_vs_q64_hand_placed.txt,
// B = 8, S = 5337, isolated, one --pmc pass): both kernels keep the matrix pipe busy 61-62 % of the time, but the chip — power-limited in
// attention — gives the 8-wave kernel 1.61 GHz and this one 1.82 GHz: half the LDS traffic (SQ_LDS_IDX_ACTIVE 185 M vs 360 M per launch)
// is the difference.  Round 2's form of it (hipcc's instruction order) kept the pipe busy 42 % of the time and lost (0.86 x); the KV loop
// below is a hand-placed instruction stream (see `body`).  1200-1208 vs 1148-1158 TFLOP/s isolated, 2410 vs 2533 us for the DiT's call with the
// fused q preparation, +0.7 % on the whole composite batch (profiles/r04_attention_q64_*.log).
// --------------------------------------------------------------------------------------------
// MFMAs of the 64-query kernel are inline asm so that the operand FILES are fixed: O accumulators and the Q fragments live
// in the AGPR half (only MFMAs touch them), the S accumulators in arch VGPRs (the softmax reads them).  With builtins hipcc
// puts S into AGPRs as well and copies 200-400 registers per KV tile back and forth (measured: 473-668 v_accvgpr / scratch
// instructions per 64 MFMAs).  hipcc neither counts nor pads what is inside an asm statement (guide 5.7):
//   * a VALU-written A / B operand needs 2 wait states before the MFMA reads it — the packed P words are a step old when their P V MFMA
//     issues, but hipcc may also RESTORE an operand it parked (v_accvgpr_read) in the instruction right before an asm statement: the step
//     statements open with two softmax instructions for that reason, and scripts/check_asm_loads.py checks every MFMA of the compiled kernel;
//   * an MFMA result needs 18 wait states (16-pass op) before anything but the next MFMA of its chain touches it: Q64_SETTLE
//     passes the registers through a nop statement before compiler-generated code may read them.
#define Q64_MFMA_S0(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b))
#define Q64_MFMA_S(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b))
#define Q64_SETTLE_S(s4) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s4[0][0]), "+v"(s4[0][1]), "+v"(s4[1][0]), "+v"(s4[1][1]))
#define Q64_SETTLE_O(o, qg) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(o[qg][0]), "+a"(o[qg][1]), "+a"(o[qg][2]), "+a"(o[qg][3]))

// pieces of the hand-placed stream (round 4) that are single statements: hipcc neither moves them against each other nor waits for them
template <int OFF>
__device__ __forceinline__ void q64_lds_read(bf16x8_t& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
// the wait that retires asm-issued fragment reads names the fragments as in/out operands: whatever the compiler does with a fragment (a
// copy to split its live range, a v_accvgpr_write to park it) then depends on the WAIT, not on the read — a copy of the read's own result
// can be scheduled straight behind the ds_read, in front of the next asm statement, before the LDS has written the registers (asm volatile
// orders asm statements, not the compiler's instructions around them)
template <int N>
__device__ __forceinline__ void q64_landed(bf16x8_t& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void q64_landed(bf16x8_t& a, bf16x8_t& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N>
__device__ __forceinline__ void q64_landed(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
__device__ __forceinline__ void q64_max3(float& d, float a, float b, float c) { asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); }
// P V: the packed P words were written >= one step (dozens of instructions) before: no VALU -> MFMA wait states to pad
#define Q64P_MFMA_O(d, a, pw) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(pw))
// A step of the stream as TWO asm statements (the LDS-DMA piece of the step, which is compiler code, sits between them).  hipcc cannot see
// into an asm statement, so whenever a statement reads a register that an earlier STATEMENT wrote and no instruction of its own has been
// issued since, its hazard recogniser pads with an s_nop (asm statements count as zero wait states): with one instruction per statement the
// softmax chain fma -> exp -> add -> cvt cost six s_nop per step, an eighth of the wave's issue slots.  Inside a statement nothing is padded
// and nothing needs to be: every v_exp result has one instruction between it and its first non-transcendental reader (the one software
// hazard of this stream on gfx950), the packed P words an MFMA reads were written a step or more before.
// softmax of two elements per query group (a: group 0, b: group 1), in place: y = s c - m c ; y = 2^y ; ps += y0 + y1 ; w = bf16x2(y0, y1)
#define Q64_V1 "v_fma_f32 %[ya0], %[sa0], %[c], %[nma]\n\t"
#define Q64_V2 "v_fma_f32 %[ya1], %[sa1], %[c], %[nma]\n\t"
#define Q64_V3 "v_exp_f32 %[ya0], %[ya0]\n\t"
#define Q64_V4 "v_fma_f32 %[yb0], %[sb0], %[c], %[nmb]\n\t"
#define Q64_V5 "v_exp_f32 %[ya1], %[ya1]\n\t"
#define Q64_V6 "v_fma_f32 %[yb1], %[sb1], %[c], %[nmb]\n\t"
#define Q64_V7 "v_add_f32 %[ta], %[ya0], %[ya1]\n\t"
#define Q64_V8 "v_exp_f32 %[yb0], %[yb0]\n\t"
#define Q64_V9 "v_add_f32 %[psa], %[psa], %[ta]\n\t"
#define Q64_V10 "v_exp_f32 %[yb1], %[yb1]\n\t"
#define Q64_V11 "v_cvt_pk_bf16_f32 %[wa], %[ya0], %[ya1]\n\t"
#define Q64_V12 "v_add_f32 %[tb], %[yb0], %[yb1]\n\t"
#define Q64_V13 "v_add_f32 %[psb], %[psb], %[tb]\n\t"
#define Q64_V14 "v_cvt_pk_bf16_f32 %[wb], %[yb0], %[yb1]"
#define Q64_SC_IN(sc, T, E0) [sa0] "v"(sc[0][T][E0]), [sa1] "v"(sc[0][T][E0 + 1]), [sb0] "v"(sc[1][T][E0]), [sb1] "v"(sc[1][T][E0 + 1])
// Every statement that holds an MFMA opens with two VALU instructions of the softmax (or more): hipcc may restore a parked operand (a K / V^T
// fragment or packed P word it kept in AGPRs across the rescale branch: v_accvgpr_read) in the instruction right before the statement, and an
// MFMA reads a VALU-written operand correctly only two wait states later — a hazard hipcc pads for its own MFMAs, not for one inside asm.
// steps 4..15, first half: wait | V1 V2 | P V (group 0) | read | V3 | P V (group 1) | read | V4-V6 | S (group 0).  C0 = "0" opens a key half
#define Q64_STEP_HI_A(C0, S0CONS, N, o0_, o1_, vf_, p0_, p1_, s0_, kf_, q0_, d1_, a1_, F1, d2_, a2_, F2, sc, T, E0, cc, nma_, nmb_)        \
  asm volatile("s_waitcnt lgkmcnt(%[n])\n\t" Q64_V1 Q64_V2                                                                                \
               "v_mfma_f32_32x32x16_bf16 %[o0], %[vf], %[p0], %[o0]\n\t"                                                                  \
               "ds_read_b128 %[d1], %[a1] offset:%[f1]\n\t" Q64_V3                                                                        \
               "v_mfma_f32_32x32x16_bf16 %[o1], %[vf], %[p1], %[o1]\n\t"                                                                  \
               "ds_read_b128 %[d2], %[a2] offset:%[f2]\n\t" Q64_V4 Q64_V5 Q64_V6                                                          \
               "v_mfma_f32_32x32x16_bf16 %[s0], %[kf], %[q0], " C0                                                                        \
               : [o0] "+a"(o0_), [o1] "+a"(o1_), [s0] S0CONS(s0_), [d1] "=&v"(d1_), [d2] "=&v"(d2_), [ya0] "=&v"(yA0), [ya1] "=&v"(yA1),  \
                 [yb0] "=&v"(yB0), [yb1] "=&v"(yB1)                                                                                       \
               : [n] "n"(N), [vf] "v"(vf_), [p0] "v"(p0_), [p1] "v"(p1_), [kf] "v"(kf_), [q0] "a"(q0_), [a1] "v"(a1_), [f1] "n"(F1),      \
                 [a2] "v"(a2_), [f2] "n"(F2), Q64_SC_IN(sc, T, E0), [c] "s"(cc), [nma] "v"(nma_), [nmb] "v"(nmb_))
// steps 4..15, second half: V7-V10 | S (group 1) | V11-V14
#define Q64_STEP_HI_B(C1, S1CONS, s1_, kf_, q1_)                                                                                          \
  asm volatile(Q64_V7 Q64_V8 Q64_V9 Q64_V10 "v_mfma_f32_32x32x16_bf16 %[s1], %[kf], %[q1], " C1 "\n\t" Q64_V11 Q64_V12 Q64_V13 Q64_V14    \
               : [s1] S1CONS(s1_), [yb0] "+v"(yB0), [yb1] "+v"(yB1), [psa] "+v"(ps[0]), [psb] "+v"(ps[1]), [wa] "=&v"(wA),                \
                 [wb] "=&v"(wB), [ta] "=&v"(sA), [tb] "=&v"(sB)                                                                           \
               : [kf] "v"(kf_), [q1] "a"(q1_), [ya0] "v"(yA0), [ya1] "v"(yA1))
// steps 0..3 (no P V yet), first half: [wait] | V1 V2 | S (group 0) | read [| read] | V3-V7 | S (group 1)
#define Q64_STEP_LO_A(WAIT, READ2, C0, C1, SCONS, N, s0_, s1_, kf_, q0_, q1_, d1_, a1_, F1, d2_, a2_, F2, sc, T, E0, cc, nma_, nmb_)       \
  asm volatile(WAIT Q64_V1 Q64_V2 "v_mfma_f32_32x32x16_bf16 %[s0], %[kf], %[q0], " C0 "\n\t"                                              \
               "ds_read_b128 %[d1], %[a1] offset:%[f1]\n\t" READ2 Q64_V3 Q64_V4 Q64_V5 Q64_V6 Q64_V7                                      \
               "v_mfma_f32_32x32x16_bf16 %[s1], %[kf], %[q1], " C1                                                                        \
               : [s0] SCONS(s0_), [s1] SCONS(s1_), [d1] "=&v"(d1_), [d2] "=&v"(d2_), [ya0] "=&v"(yA0), [ya1] "=&v"(yA1), [yb0] "=&v"(yB0),\
                 [yb1] "=&v"(yB1), [ta] "=&v"(sA)                                                                                         \
               : [n] "n"(N), [kf] "v"(kf_), [q0] "a"(q0_), [q1] "a"(q1_), [a1] "v"(a1_), [f1] "n"(F1), [a2] "v"(a2_), [f2] "n"(F2),       \
                 Q64_SC_IN(sc, T, E0), [c] "s"(cc), [nma] "v"(nma_), [nmb] "v"(nmb_))
// steps 0..3, second half: V8-V14 (the K fragment stays an operand: its registers are not the compiler's to reuse for the LDS-DMA address
// arithmetic between the two halves while the MFMA that closed the first half is a few cycles old)
#define Q64_STEP_LO_B(kf_)                                                                                                                \
  asm volatile(Q64_V8 Q64_V9 Q64_V10 Q64_V11 Q64_V12 Q64_V13 Q64_V14                                                                      \
               : [yb0] "+v"(yB0), [yb1] "+v"(yB1), [psa] "+v"(ps[0]), [psb] "+v"(ps[1]), [wa] "=&v"(wA), [wb] "=&v"(wB), [tb] "=&v"(sB)   \
               : [ya0] "v"(yA0), [ya1] "v"(yA1), [ta] "v"(sA), "v"(kf_))
template <class F, int G = 0>
__device__ __forceinline__ void q64_unroll16(F&& f) {
  if constexpr (G < 16) {
    f(std::integral_constant<int, G>{});
    q64_unroll16<F, G + 1>(f);
  }
}

#if DRAG_EXP
// experiment builds: shader-clock stamps of workgroup 0 / wave 0: [0] kernel start, [1] before the KV loop, then per tile [2 + 2 it] the wave
// reaches the tile's vmcnt(0) + barrier, [3 + 2 it] it is past them; [2 + 2 nkv] loop done, [3 + 2 nkv] kernel end
__device__ unsigned long long g_attn_stamps[512];
#define ATTN_STAMP(slot)                                                                                                  \
  do {                                                                                                                     \
    if (blockIdx.x == 0 && w == 0 && l == 0 && (slot) < 512) g_attn_stamps[(slot)] = __builtin_amdgcn_s_memtime();         \
  } while (0)
#else
#define ATTN_STAMP(slot) do { } while (0)
#endif
template <bool QPREP>
__global__ __launch_bounds__(256, 1) void attention_q64_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 K tiles | 2 V^T tiles | Q64_QLDS bytes of q rows on their way to registers
  const int w = wave_id(), l = lane_id();
  ATTN_STAMP(0);
  const int hh = l >> 5;
  constexpr int QB = 256, CPW = 4;
  const int nqb = (p.S + QB - 1) / QB;
  // An ITEM is a (batch, head) and a block of 256 queries: item & 7 = the XCD (heads 8 g + xcd share an L2), item >> 3 = (group, query block).
  // A grid of p.items workgroups runs one item each; a smaller grid (a multiple of 8: one workgroup per CU) walks items i, i + grid, ... —
  // all on the workgroup's XCD — and the KV stream of an item's last two tiles stages the NEXT item's K(0), K(1), V(0) where it would
  // stage tiles that do not exist, the next item's q rows are requested before this item's output is stored: with one workgroup per CU
  // (512 registers per wave) nothing else overlaps an item's seam — 6.5 % of an 84-tile item (profiles/r05_attn_q64_stamps.log)
  struct Item { int b, h, q0; };
  auto decode = [&](int item, Item& t) -> bool {
    const int xcd = item & 7, loc = item >> 3;
    const int bh = (loc / nqb) * 8 + xcd;
    t.b = bh / p.H;
    t.h = bh - t.b * p.H;
    t.q0 = (loc - (loc / nqb) * nqb) * QB + w * 64;              // group qg: queries q0 + 32 qg + (l & 31)
    return bh < p.B * p.H;
  };
  int item = (int)blockIdx.x;
  Item cur;
  if (!decode(item, cur)) return;      // (grids smaller than p.items are launched only when B * H is a multiple of 8: every item exists)

  bf16x8_t qf[2][8];
  auto k_descr = [&](const Item& t) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.k + (long long)t.b * p.qk_bs + t.h * 128), 0, p.k_bytes, 0x00020000);
  };
  auto v_descr = [&](const Item& t) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.vt + ((long long)(t.b * p.H + t.h) * 128) * p.s_pad), 0, p.vt_bytes, 0x00020000);
  };
  // the descriptors the LDS-DMA pieces go through, and what is subtracted from a piece's key index: in an item's last two tiles they are
  // switched to the NEXT item's K (from tile nkv - 2 on) and V^T (tile nkv - 1) with s_pad subtracted — K(nkv), K(nkv + 1), V(nkv) become
  // the next item's K(0), K(1), V(0), in the buffers where its prologue and first tile look for them (nkv even)
  __amdgpu_buffer_rsrc_t rsK = k_descr(cur), rsV = v_descr(cur);
  int ksub = 0, vsub = 0;
  // K staging offsets: lane's row of piece i inside a tile and its swizzled 16-byte slot as ONE loop-invariant byte offset; a piece's address is
  // that plus the tile's (uniform) byte offset — one v_add_u32 per piece (the clamp min(row, S - 1) and its 64-bit multiply-add cost three
  // VALU instructions per piece in a loop bound by its issue slots).  Rows past S fall outside the (batch, head) descriptor: the LDS-DMA
  // writes zeros for them, and their scores are masked by the ragged-tile path like the clamped copies of row S - 1 were.
  unsigned koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = w * CPW + i;
    const int krow = c * 4 + (l >> 4);
    koff[i] = (unsigned)krow * (unsigned)(p.ld_qk * 2) + (unsigned)(((l & 15) ^ (krow & 15)) * 16);
    const int vrow = c * 8 + (l >> 3);
    const int vslot = (l & 7) ^ ((vrow >> 1) & 7);
    voff[i] = (unsigned)(((long long)vrow * p.s_pad + vslot * 8) * 2);
  }
  auto stage_k1 = [&](int buf, int kv0, int i) {
    const int c = w * CPW + i;
    const unsigned ko = koff[i] + (unsigned)(kv0 - ksub) * (unsigned)(p.ld_qk * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (DRAG_LDS void*)((DRAG_LDS char*)smem + buf * KT_BYTES + c * 1024), 16, ko, 0, 0, 0);
  };
  auto stage_v1 = [&](int buf, int kv0, int i) {
    const int c = w * CPW + i;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (DRAG_LDS void*)((DRAG_LDS char*)smem + 2 * KT_BYTES + buf * VT_BYTES + c * 1024),
                                             16, voff[i], (kv0 - vsub) * 2, 0, 0);
  };
  const int krd = (l & 31) * 256, kx = l & 15;
  const int vrd = (l & 31) * 128, vx = ((l & 31) >> 1) & 7;
  int ak[8], av[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) ak[ks] = krd + (((2 * ks + hh) ^ kx) << 4);
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) av[s2] = vrd + (((2 * s2 + hh) ^ vx) << 4);
  // LDS byte addresses of this lane's K / V^T fragment slots: one register each, added once (inside the steps hipcc re-derived them per read)
  const unsigned lds0 = (unsigned)(size_t)(DRAG_LDS char*)smem;
  unsigned akl[8], avl[4];
  {
#pragma unroll
    for (int i = 0; i < 8; ++i) akl[i] = lds0 + (unsigned)ak[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) avl[i] = lds0 + (unsigned)av[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(akl[i]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(avl[i]));
  }

  const int nkv = p.s_pad / 64;
  const bool paired = (nkv & 1) == 0 && nkv >= 4;      // the last two tiles are peeled bodies that stage for the next item (buffer parities line up)
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto stage_first = [&]() {       // K(0), V(0), K(1) of the item rsK / rsV name
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_k1(0, 0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_v1(0, 0, i);
    if (nkv > 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) stage_k1(1, 64, i);
    }
  };
  // Q load in two parts: the rows are REQUESTED (q_request: LDS-DMA, 16 bytes x 8 per lane and query group, every lane's pieces at
  // lane-private LDS addresses — 16 KiB per wave beside the tiles, no registers) — for the next item before this item's output is
  // stored, so the round trip runs under the epilogue — and turned into fragments later (q_fragments: read back, + RMSNorm / RoPE when
  // QPREP, whose weight and table rows are L2-resident).  (Requested into registers, the 64 VGPRs were spilled around the epilogue and
  // the spill stores waited for the loads they were meant to hide.)
  // (what an item's seam derives from the lane id is recomputed per item from a laundered copy: hoisted out of the item loop it would sit
  //  in VGPRs across the KV stream, which has 38 to spare — the first build of this loop spilled 1.3 KB to scratch, inside the stream too)
  auto lane_now = [&]() { int v = l; asm volatile("" : "+v"(v)); return v; };
  DRAG_LDS char* const qlds = (DRAG_LDS char*)smem + 2 * KT_BYTES + 2 * VT_BYTES + w * 16384;
  auto q_request = [&](const Item& t, const int ll) {
    // (the K descriptor's span fits q: same row stride, same batch stride, same 128 columns of a head)
    __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.q + (long long)t.b * p.qk_bs + t.h * 128), 0, p.k_bytes, 0x00020000);
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      const int qr = min(t.q0 + 32 * qg + (ll & 31), p.S - 1);
      const unsigned vo = (unsigned)qr * (unsigned)(p.ld_qk * 2) + (unsigned)((ll >> 5) * 16);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (DRAG_LDS void*)(qlds + (qg * 8 + ks) * 1024), 16, vo, ks * 32, 0, 0);   // (the column offset as the
                                                                       // scalar offset: an instruction offset would move the LDS address as well)
    }
  };
  auto q_fragments = [&](const Item& t, const int ll) {       // (the wave's own pieces: behind its s_waitcnt vmcnt(0), no barrier)
    const int hh = ll >> 5;
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      u32x4_t raw[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) raw[ks] = *(const DRAG_LDS u32x4_t*)(qlds + (qg * 8 + ks) * 1024 + ll * 16);
      if constexpr (!QPREP) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[qg][ks] = __builtin_bit_cast(bf16x8_t, raw[ks]);
      } else {
        const int qr = min(t.q0 + 32 * qg + (ll & 31), p.S - 1);
        const bf16_t* wsel = (qr < p.s_txt ? p.wq_txt : p.wq_img) + hh * 8;
        const float* cp = p.cosT + (long long)qr * 64 + hh * 4;
        const float* sp = p.sinT + (long long)qr * 64 + hh * 4;
        u32x4_t wr[8];
        f32x4_t c4[8], s4[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          wr[ks] = *(const u32x4_t*)(wsel + ks * 16);
          c4[ks] = *(const f32x4_t*)(cp + ks * 8);
          s4[ks] = *(const f32x4_t*)(sp + ks * 8);
        }
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a0 = bf2f((bf16_t)(raw[ks][j] & 0xffff)), a1 = bf2f((bf16_t)(raw[ks][j] >> 16));
            ss += a0 * a0;
            ss += a1 * a1;
          }
        ss += __shfl_xor(ss, 32, 64);
        const float rs = rsqrtf(ss * (1.0f / 128.0f) + p.eps);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          u32x4_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a0 = rbf(rbf(bf2f((bf16_t)(raw[ks][j] & 0xffff)) * rs) * bf2f((bf16_t)(wr[ks][j] & 0xffff)));
            const float a1 = rbf(rbf(bf2f((bf16_t)(raw[ks][j] >> 16)) * rs) * bf2f((bf16_t)(wr[ks][j] >> 16)));
            o[j] = pack2bf(a0 * c4[ks][j] - a1 * s4[ks][j], a1 * c4[ks][j] + a0 * s4[ks][j]);
          }
          qf[qg][ks] = __builtin_bit_cast(bf16x8_t, o);
        }
      }
    }
  };
  stage_first();
  q_request(cur, l);               // while those tiles are in flight
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  q_fragments(cur, l);
  // From here on the Q fragments exist ONLY in the AGPR half: the tied operand makes hipcc copy each fragment into an AGPR
  // tuple once; a fragment that merely gets "a"-constrained at its uses stays in VGPRs and is re-copied before every use
  // (64 v_accvgpr_write per KV tile and 64 VGPRs gone).  The next item's fragments go there as soon as they exist (the AGPR half has
  // room beside the output accumulators; 64 more live VGPRs across the epilogue were spilled)
  bf16x8_t qa[2][8];
  auto q_to_agprs = [&]() {
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) asm volatile("; q fragment -> AGPR" : "=a"(qa[qg][ks]) : "0"(qf[qg][ks]));
  };
  q_to_agprs();
  bool staged = true;              // K(0), V(0), K(1) of the item at hand are in the LDS (or on their way)
  // The previous item's output stores are the YOUNGEST vector-memory operations when the next item begins: its q rows and first tiles were
  // requested before them, and a wave's vector-memory operations retire in issue order — so the waits of the seam leave exactly that many
  // operations in flight instead of draining the stores (their acknowledgements take microseconds).  0 = count unknown (a ragged query
  // block skips stores): wait for everything
  int stores_behind = 0;
  auto wait_older_than_stores = [&]() {
    if (stores_behind == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (stores_behind == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  for (;;) {                       // items of this workgroup
  ATTN_STAMP(4 + 2 * (p.s_pad / 64));
  const int b = cur.b, h = cur.h, q0 = cur.q0;
  const bool has_next = item + (int)gridDim.x < p.items;
  if (!staged) {                   // (an odd number of tiles: no peeled bodies staged for this item)
    __syncthreads();               // every wave is done with the previous item's tiles
    stage_first();
    stores_behind = 0;             // (these pieces are YOUNGER than the previous item's stores: the wait below has to cover everything)
  }
  f32x16_t oacc[2][4];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qg][i][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  wait_older_than_stores();                  // K(0), V(0), K(1) landed
  __syncthreads();
  f32x16_t scur[2][2], snext[2][2];          // [query group][key half]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8_t kf = *(const bf16x8_t*)(smem + ((akl[ks] - lds0) + t * (32 * 256)));
#pragma unroll
      for (int qg = 0; qg < 2; ++qg) {
        if (ks == 0) Q64_MFMA_S0(scur[qg][t], kf, qa[qg][ks]);
        else Q64_MFMA_S(scur[qg][t], kf, qa[qg][ks]);
      }
    }
  Q64_SETTLE_S(scur);
  auto mask_and_max = [&](f32x16_t (&sx)[2], const int kvs) -> float {
    if (kvs + 64 > p.S) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kvs + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= p.S) sx[t][r] = -INFINITY;
        }
    }
    float m = sx[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sx[t][r]);
    return fmaxf(m, __shfl_xor(m, 32, 64));
  };
  float mt_carry[2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) mt_carry[qg] = mask_and_max(scur[qg], 0);

  // ---- round 4: the KV loop as a HAND-PLACED instruction stream.  One wave per SIMD issues in order: two MFMAs back to back stall the wave
  // for the 28 cycles the first one still occupies the pipe, and nothing behind them issues — the round-2 form of this loop (hipcc's order
  // between sched_barriers: PV PV reads QK QK, then the step's 18 VALU as one block) kept the pipe busy 42 % of the time (PMC).  Here every
  // MFMA is followed by its share of the step's other work — at most 5 single-issue instructions per gap, what a lone wave hides behind
  // a 32-cycle MFMA: the step's 14 softmax VALU in a hazard-free order (a transcendental's input / result is never produced / consumed by
  // the neighbouring instruction), the two fragment reads of step g + 2 (the reads run TWO steps ahead: nobody else covers a lone wave's
  // LDS latency) and one LDS-DMA piece.  Everything is `asm volatile`: hipcc keeps the order and inserts no waits; lgkmcnt is counted by hand.
  // Same arithmetic per query group as before: same bits as the 8-wave kernel.
  // The last key group's P V (8 MFMAs on fragments and P words already in registers) and the row maxima of the tile that follows are
  // one block at the TOP of the next iteration: behind the barrier, with the first K fragment reads of the new tile already issued, the
  // eight MFMAs cover both the reads' latency and the 32 v_max3 of the two row-maximum trees (4 per gap).
  bf16x8_t vtr[4];              // the four V^T fragments of the trailing P V MFMAs (read in steps 14 / 15, used after the next barrier)
  u32x4_t pk[2][4];             // packed P: key group kg of a tile is consumed in its steps 4 kg + 4 .. 4 kg + 7, group 3 after the next barrier
  // knext / vnext: the K / V^T pieces of this tile's stream are the NEXT item's (the item's last two tiles, whose own pieces would be tiles
  // that do not exist): K(0) into buffer 0 from tile nkv - 2, K(1) into buffer 1 and V(0) into buffer 0 from tile nkv - 1
  auto body = [&](const int it, auto par, f32x16_t (&sc)[2][2], f32x16_t (&sn)[2][2], auto firstc) {
    constexpr int PAR = decltype(par)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    if constexpr (!FIRST) {
      if (paired && it >= nkv - 2) {                 // (two scalar compares per tile; the switch itself once per item)
        Item nx;
        const bool have = has_next && decode(item + (int)gridDim.x, nx);
        if (PAR == 0) {
          rsK = have ? k_descr(nx) : __builtin_amdgcn_make_buffer_rsrc((void*)p.k, 0, 0, 0x00020000);    // (no next item: zero records, the pieces stage zeros)
          ksub = p.s_pad;
        } else {
          rsV = have ? v_descr(nx) : __builtin_amdgcn_make_buffer_rsrc((void*)p.vt, 0, 0, 0x00020000);
          vsub = p.s_pad;
        }
      }
    }
    constexpr int KN = ((PAR + 1) & 1) * KT_BYTES;
    constexpr int VB = 2 * KT_BYTES + PAR * VT_BYTES;
    const int kv0 = it * 64;
#if DRAG_EXP
    const unsigned long long t_reach = __builtin_amdgcn_s_memtime();      // (stored behind the barrier: a store here would be waited for by the vmcnt(0))
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#if DRAG_EXP
    if (blockIdx.x == 0 && w == 0 && l == 0 && 3 + 2 * it < 512) {
      g_attn_stamps[2 + 2 * it] = t_reach;
      g_attn_stamps[3 + 2 * it] = __builtin_amdgcn_s_memtime();
    }
#endif
    bf16x8_t kfr[3], vfr[3];      // fragment rings (step g uses slot g % 3)
    // RULE of this kernel: a fragment requested by an asm ds_read reaches compiler-visible code only through q64_landed (the s_waitcnt
    // that retires it, with the fragment as an in/out operand), and between the read and that wait there are asm statements only.  hipcc
    // takes the read's destination for defined the moment the statement ends: a build that kept K(0) / K(1) in flight across the rescale
    // branch below had them parked in AGPRs one instruction after the request (v_accvgpr_write of registers the LDS had not written yet:
    // NaNs that came and went with register allocation).  scripts/check_asm_loads.py walks the compiled kernel for exactly that
    // (tests/test_asm_load_discipline.py).
    // K(0), K(1) of the next tile first: their latency hides behind the trailing P V block, which is asm only; they have landed before
    // the compiler's code (maxima across lane halves, rescale decision) begins
    q64_lds_read<KN>(kfr[0], akl[0]);
    q64_lds_read<KN>(kfr[1], akl[1]);
    if constexpr (FIRST) q64_landed<0>(kfr[0], kfr[1]);
    if constexpr (!FIRST) {
      // row maxima of sc (32 scores per lane and query group) as two v_max3 trees: 10 + 4 + 2 instructions each, group A and B interleaved
      float ta[10], tb[10], ua[4], ub[4], ra, rb;
      auto el = [&](int qg, int i) -> float { return sc[qg][i >> 4][i & 15]; };
      auto L1 = [&](float* t, int qg, int i) { q64_max3(t[i], el(qg, 3 * i), el(qg, 3 * i + 1), el(qg, 3 * i + 2)); };
      auto L2 = [&](float* u, const float* t, int qg, int i) {
        if (i < 3) q64_max3(u[i], t[3 * i], t[3 * i + 1], t[3 * i + 2]);
        else q64_max3(u[3], t[9], el(qg, 30), el(qg, 31));
      };
#define Q64P_T(dt, qg) Q64P_MFMA_O(oacc[qg][dt], vtr[dt], pk[qg][3])
      // six v_max3 behind each of the first five MFMAs (what a lone wave hides behind 32 cycles); the joining of the lane halves behind the
      // last three: the other 32 keys of a query live in lane ^ 32, v_permlane32_swap pairs the halves of both groups (no LDS round trip: a
      // ds_bpermute here waits out the K fragment reads already in flight).  swap(ra, rb) = ((ra.lo | rb.lo), (ra.hi | rb.hi)): lanes < 32 then
      // hold group A's two halves of query l, lanes >= 32 group B's of query l - 32; their maximum, swapped with itself, is group A's value in
      // both halves of the first result and group B's in the second.  (The swap needs two wait states after the VALU write of its operands:
      // the MFMA between and one s_nop.)
      // As asm: hipcc (ROCm 7.2) folds fmaxf(sw[0], sw[1]) of __builtin_amdgcn_permlane32_swap's two results to sw[0], and the two results
      // of swap(x, x) to one value (the optimised IR holds a single extractvalue): the maxima then covered the keys of ONE lane half — every
      // parity test passed (any reference value below the true maximum gives the same softmax until 2^(max - reference) overflows) and a key 128
      // octaves above its row's running maximum gave NaN.  scripts/probe/dbg_attn_rescale*.py; test_attention_hot_key_in_every_lane_half.
      Q64P_T(0, 0); L1(ta, 0, 0); L1(ta, 0, 1); L1(ta, 0, 2); L1(ta, 0, 3); L1(ta, 0, 4); L1(ta, 0, 5);
      Q64P_T(0, 1); L1(ta, 0, 6); L1(ta, 0, 7); L1(ta, 0, 8); L1(ta, 0, 9); L1(tb, 1, 0); L1(tb, 1, 1);
      Q64P_T(1, 0); L1(tb, 1, 2); L1(tb, 1, 3); L1(tb, 1, 4); L1(tb, 1, 5); L1(tb, 1, 6); L1(tb, 1, 7);
      Q64P_T(1, 1); L1(tb, 1, 8); L1(tb, 1, 9); L2(ua, ta, 0, 0); L2(ua, ta, 0, 1); L2(ua, ta, 0, 2); L2(ua, ta, 0, 3);
      Q64P_T(2, 0); L2(ub, tb, 1, 0); L2(ub, tb, 1, 1); L2(ub, tb, 1, 2); L2(ub, tb, 1, 3); q64_max3(ra, ua[0], ua[1], ua[2]); q64_max3(rb, ub[0], ub[1], ub[2]);
      asm volatile("v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3" : "+v"(ra), "+v"(rb) : "v"(ua[3]), "v"(ub[3]));
      Q64P_T(2, 1);
      asm volatile("s_nop 0\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0" : "+v"(ra), "+v"(rb));
      Q64P_T(3, 0);
      asm volatile("s_nop 0\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ra), "+v"(rb));
      Q64P_T(3, 1);
#undef Q64P_T
      q64_landed<0>(kfr[0], kfr[1]);
      mt_carry[0] = ra;                                     // group A's maximum in both lane halves
      mt_carry[1] = rb;                                     // group B's
      if (kv0 + 64 > p.S) {       // the ragged last tile: keys >= S do not exist (the trees above saw them) — once per (batch, head, query block)
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) mt_carry[qg] = mask_and_max(sc[qg], kv0);
      }
    }
    float nmc[2];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      const float mt = mt_carry[qg];
      if (!__all((mt - m_run[qg]) * p.c <= DEFER_THR)) {
        Q64_SETTLE_O(oacc, qg);
        const float m_new = fmaxf(m_run[qg], mt);
        const float alpha = __builtin_amdgcn_exp2f((m_run[qg] - m_new) * p.c);
        l_run[qg] *= alpha;
        m_run[qg] = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[qg][i][r] *= alpha;
      }
      nmc[qg] = -(m_run[qg] * p.c);
    }
    float ps[2] = {0.f, 0.f};
    // one step; G is a compile-time constant (ring slots, immediate offsets, which pieces exist)
    auto step = [&](auto gc) {
      constexpr int G = decltype(gc)::value;
      constexpr int T = G >> 3, E0 = (2 * G) & 15;
      // reads issued in step s (in order): K(s+2) if s+2 <= 15; V(s+2) if 4 <= s+2 <= 15; the 4 trailing fragments in steps 14 / 15
      constexpr int prev = G - 1;
      // the step's wait: K(G) (and V(G)) landed; step G-1's reads stay in flight
      constexpr int younger = prev < 0 ? 0 : ((prev + 2 <= 15) + (prev + 2 >= 4 && prev + 2 <= 15) + (prev >= 14 ? 2 : 0));
      float yA0, yA1, yB0, yB1, sA, sB;
      uint32_t wA, wB;
      auto dma = [&]() { if constexpr (G < CPW) stage_k1(PAR, kv0 + 128, G); else if constexpr (G < 2 * CPW) stage_v1((PAR + 1) & 1, kv0 + 64, G - CPW); };
      constexpr int KOFF = KN + ((G + 2) >> 3) * (32 * 256), VOFF = VB + ((G + 2) & 3) * (32 * 128);
      if constexpr (G >= 14) {            // P V + S; the reads are the trailing V^T fragments (two per step)
        constexpr int F1 = VB + (2 * (G - 14)) * (32 * 128), F2 = F1 + 32 * 128;
        Q64_STEP_HI_A("%[s0]", "+v", younger, oacc[0][G & 3], oacc[1][G & 3], vfr[G % 3], pk[0][(G >> 2) - 1], pk[1][(G >> 2) - 1], sn[0][T],
                      kfr[G % 3], qa[0][G & 7], vtr[2 * (G - 14)], avl[3], F1, vtr[2 * (G - 14) + 1], avl[3], F2, sc, T, E0, p.c, nmc[0], nmc[1]);
        dma();
        Q64_STEP_HI_B("%[s1]", "+v", sn[1][T], kfr[G % 3], qa[1][G & 7]);
      } else if constexpr (G == 8) {      // opens the second key half of S: C = 0
        Q64_STEP_HI_A("0", "=&v", younger, oacc[0][G & 3], oacc[1][G & 3], vfr[G % 3], pk[0][(G >> 2) - 1], pk[1][(G >> 2) - 1], sn[0][T],
                      kfr[G % 3], qa[0][G & 7], kfr[(G + 2) % 3], akl[(G + 2) & 7], KOFF, vfr[(G + 2) % 3], avl[((G + 2) >> 2) - 1], VOFF, sc, T, E0,
                      p.c, nmc[0], nmc[1]);
        dma();
        Q64_STEP_HI_B("0", "=&v", sn[1][T], kfr[G % 3], qa[1][G & 7]);
      } else if constexpr (G >= 4) {
        Q64_STEP_HI_A("%[s0]", "+v", younger, oacc[0][G & 3], oacc[1][G & 3], vfr[G % 3], pk[0][(G >> 2) - 1], pk[1][(G >> 2) - 1], sn[0][T],
                      kfr[G % 3], qa[0][G & 7], kfr[(G + 2) % 3], akl[(G + 2) & 7], KOFF, vfr[(G + 2) % 3], avl[((G + 2) >> 2) - 1], VOFF, sc, T, E0,
                      p.c, nmc[0], nmc[1]);
        dma();
        Q64_STEP_HI_B("%[s1]", "+v", sn[1][T], kfr[G % 3], qa[1][G & 7]);
      } else if constexpr (G == 0) {      // K(0) landed at the top of the body; opens the first key half: C = 0; one read (K(2))
        bf16x8_t none;
        Q64_STEP_LO_A("", "", "0", "0", "=&v", 0, sn[0][T], sn[1][T], kfr[0], qa[0][0], qa[1][0], kfr[2], akl[2], KOFF, none, akl[2], 0, sc, T, E0,
                      p.c, nmc[0], nmc[1]);
        dma();
        Q64_STEP_LO_B(kfr[0]);
      } else if constexpr (G == 1) {      // one read (K(3))
        bf16x8_t none;
        Q64_STEP_LO_A("s_waitcnt lgkmcnt(%[n])\n\t", "", "%[s0]", "%[s1]", "+v", younger, sn[0][T], sn[1][T], kfr[G % 3], qa[0][G], qa[1][G],
                      kfr[(G + 2) % 3], akl[G + 2], KOFF, none, akl[2], 0, sc, T, E0, p.c, nmc[0], nmc[1]);
        dma();
        Q64_STEP_LO_B(kfr[G % 3]);
      } else {                            // G = 2, 3: K(G+2) and V(G+2)
        Q64_STEP_LO_A("s_waitcnt lgkmcnt(%[n])\n\t", "ds_read_b128 %[d2], %[a2] offset:%[f2]\n\t", "%[s0]", "%[s1]", "+v", younger, sn[0][T],
                      sn[1][T], kfr[G % 3], qa[0][G], qa[1][G], kfr[(G + 2) % 3], akl[G + 2], KOFF, vfr[(G + 2) % 3], avl[((G + 2) >> 2) - 1], VOFF,
                      sc, T, E0, p.c, nmc[0], nmc[1]);
        dma();
        Q64_STEP_LO_B(kfr[G % 3]);
      }
      pk[0][G >> 2][G & 3] = wA;
      pk[1][G >> 2][G & 3] = wB;
    };
    q64_unroll16(step);
    q64_landed<0>(vtr[0], vtr[1], vtr[2], vtr[3]);              // the trailing fragments landed: every read of this tile's buffers is complete
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) l_run[qg] += ps[qg];
  };
  {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    ATTN_STAMP(1);
    body(0, P0{}, scur, snext, std::true_type{});                // (the first tile's row maxima come from the prologue)
    if (nkv > 1) body(1, P1{}, snext, scur, std::false_type{});
    for (int it = 2; it < nkv; it += 2) {
      body(it, P0{}, scur, snext, std::false_type{});
      if (it + 1 < nkv) body(it + 1, P1{}, snext, scur, std::false_type{});
    }
    // the last tile's last key group
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int qg = 0; qg < 2; ++qg) Q64P_MFMA_O(oacc[qg][dt], vtr[dt], pk[qg][3]);
    ATTN_STAMP(2 + 2 * nkv);
  }
  const int le = lane_now();         // the epilogue's and the next item's lane arithmetic (see lane_now)
  Item nxt = cur;
  if (has_next) {                    // the next item's q rows: requested before this item's output is stored, in flight under the epilogue
    decode(item + (int)gridDim.x, nxt);
    q_request(nxt, le);
  }

#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    Q64_SETTLE_O(oacc, qg);
    const float lt = l_run[qg] + __shfl_xor(l_run[qg], 32, 64);
    const float inv = 1.0f / lt;
    const int qrow = q0 + 32 * qg + (le & 31);
    if (p.tune & 2) {
      bf16_t* op = p.out + (long long)b * p.o_bs + (long long)qrow * p.ld_o + h * 128 + 8 * (le >> 5);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          uint32_t a0 = pack2bf(oacc[qg][dt][4 * g] * inv, oacc[qg][dt][4 * g + 1] * inv);
          uint32_t a1 = pack2bf(oacc[qg][dt][4 * g + 2] * inv, oacc[qg][dt][4 * g + 3] * inv);
          uint32_t b0 = pack2bf(oacc[qg][dt][4 * g + 4] * inv, oacc[qg][dt][4 * g + 5] * inv);
          uint32_t b1 = pack2bf(oacc[qg][dt][4 * g + 6] * inv, oacc[qg][dt][4 * g + 7] * inv);
          const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
          if (qrow < p.S) *(u32x4_t*)(op + 32 * dt + 8 * g) = o;
        }
    } else if (qrow < p.S) {
      bf16_t* op = p.out + (long long)b * p.o_bs + (long long)qrow * p.ld_o + h * 128 + 4 * (le >> 5);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2_t o;
          o[0] = pack2bf(oacc[qg][dt][4 * g] * inv, oacc[qg][dt][4 * g + 1] * inv);
          o[1] = pack2bf(oacc[qg][dt][4 * g + 2] * inv, oacc[qg][dt][4 * g + 3] * inv);
          *(u32x2_t*)(op + 32 * dt + 8 * g) = o;
        }
    }
  }
  ATTN_STAMP(3 + 2 * (p.s_pad / 64));
  if (!has_next) break;
  stores_behind = q0 + 64 <= p.S ? ((p.tune & 2) ? 16 : 32) : 0;      // (every row of the wave valid: every store instruction was issued)
  wait_older_than_stores();        // the q rows (the RoPE table rows q_fragments loads queue behind the stores; Q fragments computed BEFORE the
                                   //  epilogue instead — rows requested a tile earlier — cost spills in the q-preparation instantiation)
  q_fragments(nxt, le);
  q_to_agprs();
  ATTN_STAMP(5 + 2 * (p.s_pad / 64));
  item += (int)gridDim.x;
  cur = nxt;
  if (!paired) {                   // (paired: the stream switched both already)
    rsK = k_descr(cur);
    rsV = v_descr(cur);
  }
  ksub = vsub = 0;
  staged = paired;
  }
}


// --------------------------------------------------------------------------------------------
// attention_q64g_kernel (round 6) — attention_q64_kernel's walking form with the WHOLE KV loop of an item as ONE generated asm statement
// (scripts/gen/attn_q64_tile.py -> attn_q64_tile.h: prologue scores, every tile with its barrier, maxima, rescale decision and staging
// pieces, the final P V), operands bound to the physical registers the text names.  hipcc's part: the item walk, the q rows / fragments,
// the staging of an item's first tiles when nobody staged them, the epilogue.  What the generated form buys:
// no compiler code inside a tile (round 5 counted ~60 instructions and 19 s_nop per tile between ~40 statements), the K / V^T fragments
// in AGPRs (only ds_read and MFMA touch them: 40 VGPRs free), and with those the FOLD form: scale * log2(e) goes into the q preparation's one
// rounding and -M (the running maximum in log2 units) into the C operand of each score chain's first MFMA, so the scores ARE the
// exponents and the 64 v_fma per tile go (349 instead of 421 instructions per tile and wave; the old stream: 425 + ~60).
// FOLD = false: every float operation and every accumulator's MFMA order of attention_q64_kernel: same bits
// (test_attention_q64_generated_stream_*).  FOLD = true (QPREP only: the fold needs the q preparation's rounding point): differs by
// design — q carries one rounding of q c instead of q; bars of the oracle tests.
// Launched for an even number >= 4 of KV tiles (the stream's last two tiles stage the next item's first tiles); otherwise, and under
// "attn_gen" = 1, attention_q64_kernel runs.
// --------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(32))) float aq_f32x32_t;
typedef __attribute__((ext_vector_type(8))) uint32_t aq_u32x8_t;
typedef __attribute__((ext_vector_type(16))) uint32_t aq_u32x16_t;

template <bool QPREP, bool FOLD, int VAR = 0>      // VAR > 0: schedule variants of the fold form (DRAG_EXPERIMENTS builds: "attn_gen" = 10 + VAR)
__global__ __launch_bounds__(256, 1) void attention_q64g_kernel(AttnArgs p) {
  static_assert(QPREP || !FOLD, "the fold needs the q preparation");
  static_assert(VAR == 0 || (FOLD && DRAG_EXP), "variants: fold form, experiment builds");
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 K tiles | 2 V^T tiles | 4 x 16 KiB of q rows on their way to registers
  const int w = wave_id(), l = lane_id();
  constexpr int QB = 256, CPW = 4;
  const int nqb = (p.S + QB - 1) / QB;
  struct Item { int b, h, q0; };
  auto decode = [&](int item, Item& t) -> bool {
    const int xcd = item & 7, loc = item >> 3;
    const int bh = (loc / nqb) * 8 + xcd;
    t.b = bh / p.H;
    t.h = bh - t.b * p.H;
    t.q0 = (loc - (loc / nqb) * nqb) * QB + w * 64;
    return bh < p.B * p.H;
  };
  int item = (int)blockIdx.x;
  Item cur;
  if (!decode(item, cur)) return;
  auto k_descr = [&](const Item& t) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.k + (long long)t.b * p.qk_bs + t.h * 128), 0, p.k_bytes, 0x00020000);
  };
  auto v_descr = [&](const Item& t) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.vt + ((long long)(t.b * p.H + t.h) * 128) * p.s_pad), 0, p.vt_bytes, 0x00020000);
  };
  auto lane_now = [&]() { int v = l; asm volatile("" : "+v"(v)); return v; };
  // per-lane constants of the stream, recomputed per item from a laundered lane id (every VGPR but v[240:255] belongs to the statement: held
  // across it they would be spilled): [0:7] K fragment LDS addresses per k-slice | [8:11] V^T fragment addresses per key group,
  // [12:15] / [16:19] global byte offsets of this wave's four K / V^T staging pieces
  const unsigned lds0 = (unsigned)(size_t)(DRAG_LDS char*)smem;
  auto lane_consts = [&](int ll, aq_u32x8_t& akl, aq_u32x16_t& misc) {
    const int hh = ll >> 5;
    const int krd = (ll & 31) * 256, kx = ll & 15;
    const int vrd = (ll & 31) * 128, vx = ((ll & 31) >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) akl[ks] = lds0 + (unsigned)(krd + (((2 * ks + hh) ^ kx) << 4));
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) misc[s2] = lds0 + (unsigned)(vrd + (((2 * s2 + hh) ^ vx) << 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = w * CPW + i;
      const int krow = c * 4 + (ll >> 4);
      misc[4 + i] = (unsigned)krow * (unsigned)(p.ld_qk * 2) + (unsigned)(((ll & 15) ^ (krow & 15)) * 16);
      const int vrow = c * 8 + (ll >> 3);
      const int vslot = (ll & 7) ^ ((vrow >> 1) & 7);
      misc[8 + i] = (unsigned)(((long long)vrow * p.s_pad + vslot * 8) * 2);
      misc[12 + i] = 0u;
    }
  };
  auto stage_first = [&](const Item& t, int ll) {       // K(0), V^T(0), K(1) of item t (compiler code: once per workgroup)
    __amdgpu_buffer_rsrc_t rsK = k_descr(t), rsV = v_descr(t);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = w * CPW + i;
      const int krow = c * 4 + (ll >> 4);
      const unsigned ko = (unsigned)krow * (unsigned)(p.ld_qk * 2) + (unsigned)(((ll & 15) ^ (krow & 15)) * 16);
      const int vrow = c * 8 + (ll >> 3);
      const int vslot = (ll & 7) ^ ((vrow >> 1) & 7);
      const unsigned vo = (unsigned)(((long long)vrow * p.s_pad + vslot * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (DRAG_LDS void*)((DRAG_LDS char*)smem + c * 1024), 16, ko, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (DRAG_LDS void*)((DRAG_LDS char*)smem + 2 * KT_BYTES + c * 1024), 16, vo, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (DRAG_LDS void*)((DRAG_LDS char*)smem + KT_BYTES + c * 1024), 16,
                                               ko + 64u * (unsigned)(p.ld_qk * 2), 0, 0, 0);
    }
  };
  DRAG_LDS char* const qlds = (DRAG_LDS char*)smem + 2 * KT_BYTES + 2 * VT_BYTES + w * 16384;
  auto q_request = [&](const Item& t, const int ll) {
    __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.q + (long long)t.b * p.qk_bs + t.h * 128), 0, p.k_bytes, 0x00020000);
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      const int qr = min(t.q0 + 32 * qg + (ll & 31), p.S - 1);
      const unsigned vo = (unsigned)qr * (unsigned)(p.ld_qk * 2) + (unsigned)((ll >> 5) * 16);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (DRAG_LDS void*)(qlds + (qg * 8 + ks) * 1024), 16, vo, ks * 32, 0, 0);
    }
  };
  // the Q fragments as the statement takes them: two 32-register AGPR tuples, [group][k-slice][4]
  auto q_fragments = [&](const Item& t, const int ll, aq_f32x32_t (&qin)[2]) {
    const int hh = ll >> 5;
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      u32x4_t raw[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) raw[ks] = *(const DRAG_LDS u32x4_t*)(qlds + (qg * 8 + ks) * 1024 + ll * 16);
      if constexpr (!QPREP) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t word = raw[ks][j];      // (a copy first: __builtin_bit_cast applied to the vector ELEMENT expression itself was compiled
            qin[qg][4 * ks + j] = __builtin_bit_cast(float, word);      //  as a splat of element 0 — hipcc 7.2; found by the bit comparison on the GPU)
          }
      } else {
        const int qr = min(t.q0 + 32 * qg + (ll & 31), p.S - 1);
        const bf16_t* wsel = (qr < p.s_txt ? p.wq_txt : p.wq_img) + hh * 8;
        const float* cp = p.cosT + (long long)qr * 64 + hh * 4;
        const float* sp = p.sinT + (long long)qr * 64 + hh * 4;
        u32x4_t wr[8];
        f32x4_t c4[8], s4[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          wr[ks] = *(const u32x4_t*)(wsel + ks * 16);
          c4[ks] = *(const f32x4_t*)(cp + ks * 8);
          s4[ks] = *(const f32x4_t*)(sp + ks * 8);
        }
        float ss = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a0 = bf2f((bf16_t)(raw[ks][j] & 0xffff)), a1 = bf2f((bf16_t)(raw[ks][j] >> 16));
            ss += a0 * a0;
            ss += a1 * a1;
          }
        ss += __shfl_xor(ss, 32, 64);
        const float rs = rsqrtf(ss * (1.0f / 128.0f) + p.eps);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a0 = rbf(rbf(bf2f((bf16_t)(raw[ks][j] & 0xffff)) * rs) * bf2f((bf16_t)(wr[ks][j] & 0xffff)));
            const float a1 = rbf(rbf(bf2f((bf16_t)(raw[ks][j] >> 16)) * rs) * bf2f((bf16_t)(wr[ks][j] >> 16)));
            float r0 = a0 * c4[ks][j] - a1 * s4[ks][j], r1 = a1 * c4[ks][j] + a0 * s4[ks][j];
            if constexpr (FOLD) { r0 *= p.c; r1 *= p.c; }        // scale * log2(e) inside the rotation's one rounding
            qin[qg][4 * ks + j] = __builtin_bit_cast(float, pack2bf(r0, r1));
          }
      }
    }
  };
  const int nkv = p.s_pad / 64;
  aq_f32x32_t qin[2];
  stage_first(cur, l);
  q_request(cur, l);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  q_fragments(cur, l, qin);
  int stores_behind = 0;
  auto wait_older_than_stores = [&]() {
    if (stores_behind == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (stores_behind == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  for (;;) {                       // items of this workgroup
    const int b = cur.b, h = cur.h, q0 = cur.q0;
    const bool has_next = item + (int)gridDim.x < p.items;
    Item nxt = cur;
    const bool have = has_next && decode(item + (int)gridDim.x, nxt);
    const __amdgpu_buffer_rsrc_t rsK = k_descr(cur), rsV = v_descr(cur);
    // no next item: zero records, the stream's last pieces stage zeros
    const __amdgpu_buffer_rsrc_t rsKn = have ? k_descr(nxt) : __builtin_amdgcn_make_buffer_rsrc((void*)p.k, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVn = have ? v_descr(nxt) : __builtin_amdgcn_make_buffer_rsrc((void*)p.vt, 0, 0, 0x00020000);
    aq_u32x8_t akl;
    aq_u32x16_t misc;
    lane_consts(lane_now(), akl, misc);
    const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)w * 4096u));
    const unsigned npairs = (unsigned)(nkv - 2) / 2u;
    const unsigned ktile = 64u * (unsigned)(p.ld_qk * 2);
    const unsigned nvalid = (unsigned)(p.S - (nkv - 1) * 64);
    wait_older_than_stores();                  // K(0), V^T(0), K(1) of this item landed (this wave's pieces)
    __syncthreads();
    aq_f32x32_t o4[4];
    f32x2_t lrun;
    [[maybe_unused]] f32x16_t lacc[2];
#define AQ64_OUTS "={a[0:31]}"(o4[0]), "={a[32:63]}"(o4[1]), "={a[64:95]}"(o4[2]), "={a[96:127]}"(o4[3]), "={v[208:209]}"(lrun)
#define AQ64_INS "{a[128:159]}"(qin[0]), "{a[160:191]}"(qin[1]), "{v[176:183]}"(akl), "{v[184:199]}"(misc), [rsk] "s"(rsK), [rsv] "s"(rsV), \
                 [rskn] "s"(rsKn), [rsvn] "s"(rsVn), [npairs] "s"(npairs), [ktile] "s"(ktile), [nvalid] "s"(nvalid), [ldsw] "s"(ldsw)
    if constexpr (!FOLD) asm volatile(AQ64_ITEM_NOFOLD : AQ64_OUTS : AQ64_INS, [c] "s"(p.c) : AQ64_CLOBBERS);
    else if constexpr (VAR == 0) asm volatile(AQ64_ITEM_FOLD : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
#if DRAG_EXP
    else if constexpr (VAR == 1) asm volatile(AQ64_ITEM_FOLD_V1 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 2) asm volatile(AQ64_ITEM_FOLD_V2 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 3) asm volatile(AQ64_ITEM_FOLD_V3 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 4) asm volatile(AQ64_ITEM_FOLD_V4 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 5) asm volatile(AQ64_ITEM_FOLD_V5 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 6) asm volatile(AQ64_ITEM_FOLD_V6 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 7) asm volatile(AQ64_ITEM_FOLD_V7 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 8) asm volatile(AQ64_ITEM_FOLD_V8 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 9) asm volatile(AQ64_ITEM_FOLD_V9 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 10) asm volatile(AQ64_ITEM_FOLD_V10 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 11) asm volatile(AQ64_ITEM_FOLD_V11 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 12) asm volatile(AQ64_ITEM_FOLD_V12 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 13) asm volatile(AQ64_ITEM_FOLD_V13 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 14) asm volatile(AQ64_ITEM_FOLD_V14 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 15) asm volatile(AQ64_ITEM_FOLD_V15 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 16) asm volatile(AQ64_ITEM_FOLD_V16 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 17) asm volatile(AQ64_ITEM_FOLD_V17 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 19) asm volatile(AQ64_ITEM_FOLD_V19 : AQ64_OUTS : AQ64_INS : AQ64_CLOBBERS);
    else if constexpr (VAR == 18)         // row sums on the matrix core: two more accumulator tuples come back instead of the l registers
      asm volatile(AQ64_ITEM_FOLD_V18
                   : "={a[0:31]}"(o4[0]), "={a[32:63]}"(o4[1]), "={a[64:95]}"(o4[2]), "={a[96:127]}"(o4[3]), "={a[224:239]}"(lacc[0]), "={a[240:255]}"(lacc[1])
                   : AQ64_INS : AQ64_CLOBBERS_LSUM);
#endif
#undef AQ64_OUTS
#undef AQ64_INS
    const int le = lane_now();
    if (has_next) q_request(nxt, le);          // the next item's q rows: in flight under the epilogue
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      float lt;
      if constexpr (VAR == 18) lt = lacc[qg][0];          // (the whole row's sum, in both lane halves)
      else lt = lrun[qg] + __shfl_xor(lrun[qg], 32, 64);
      const float inv = 1.0f / lt;
      const int qrow = q0 + 32 * qg + (le & 31);
      auto oa = [&](int dt, int r) -> float { return o4[2 * qg + (dt >> 1)][16 * (dt & 1) + r]; };
      if (p.tune & 2) {
        bf16_t* op = p.out + (long long)b * p.o_bs + (long long)qrow * p.ld_o + h * 128 + 8 * (le >> 5);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            uint32_t a0 = pack2bf(oa(dt, 4 * g) * inv, oa(dt, 4 * g + 1) * inv);
            uint32_t a1 = pack2bf(oa(dt, 4 * g + 2) * inv, oa(dt, 4 * g + 3) * inv);
            uint32_t b0 = pack2bf(oa(dt, 4 * g + 4) * inv, oa(dt, 4 * g + 5) * inv);
            uint32_t b1 = pack2bf(oa(dt, 4 * g + 6) * inv, oa(dt, 4 * g + 7) * inv);
            const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            const u32x4_t o = {r0[0], r1[0], r0[1], r1[1]};
            if (qrow < p.S) *(u32x4_t*)(op + 32 * dt + 8 * g) = o;
          }
      } else if (qrow < p.S) {
        bf16_t* op = p.out + (long long)b * p.o_bs + (long long)qrow * p.ld_o + h * 128 + 4 * (le >> 5);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            u32x2_t o;
            o[0] = pack2bf(oa(dt, 4 * g) * inv, oa(dt, 4 * g + 1) * inv);
            o[1] = pack2bf(oa(dt, 4 * g + 2) * inv, oa(dt, 4 * g + 3) * inv);
            *(u32x2_t*)(op + 32 * dt + 8 * g) = o;
          }
      }
    }
    if (!has_next) break;
    stores_behind = q0 + 64 <= p.S ? ((p.tune & 2) ? 16 : 32) : 0;
    wait_older_than_stores();        // the next item's q rows (requested before the stores; its first tiles were staged by the stream)
    q_fragments(nxt, le, qin);
    item += (int)gridDim.x;
    cur = nxt;
  }
}

}  // namespace

#if DRAG_EXP
extern "C" int drag_debug_attn_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int drag_qk_norm_rope_vt_bf16(void* qkv, void* vt, const void* wq_txt, const void* wk_txt,
                                         const void* wq_img, const void* wk_img, const float* rope_cos,
                                         const float* rope_sin, int32_t B, int32_t S, int32_t H, int32_t ld,
                                         int32_t s_txt, float eps, void* stream) {
  DRAG_CHECK(qkv && vt, "drag_qk_norm_rope_vt_bf16: null pointer");
  DRAG_CHECK((wq_txt && wk_txt && wq_img && wk_img) || (!wq_txt && !wk_txt && !wq_img && !wk_img),
             "drag_qk_norm_rope_vt_bf16: norm weights are all-or-none");
  DRAG_CHECK((rope_cos == nullptr) == (rope_sin == nullptr), "drag_qk_norm_rope_vt_bf16: cos/sin come in pairs");
  DRAG_CHECK(B > 0 && S > 0 && H > 0 && ld >= 3 * H * 128 && ld % 8 == 0, "drag_qk_norm_rope_vt_bf16: bad shape");
  PrepArgs p;
  p.qkv = (bf16_t*)qkv; p.vt = (bf16_t*)vt;
  p.wq_txt = (const bf16_t*)wq_txt; p.wk_txt = (const bf16_t*)wk_txt;
  p.wq_img = (const bf16_t*)wq_img; p.wk_img = (const bf16_t*)wk_img;
  p.cosT = rope_cos; p.sinT = rope_sin;
  p.B = B; p.S = S; p.H = H; p.ld = ld; p.s_txt = s_txt; p.s_pad = (S + 63) / 64 * 64; p.eps = eps;
  p.skip_q = 0;
  hipLaunchKernelGGL(qk_norm_rope_vt_kernel, dim3((S + 63) / 64, H, B), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

// k and v only: the q third of the pass (a read + a write of M x D) moves into the attention kernel's fragment load
extern "C" int drag_k_norm_rope_vt_bf16(void* qkv, void* vt, const void* wk_txt, const void* wk_img, const float* rope_cos,
                                        const float* rope_sin, int32_t B, int32_t S, int32_t H, int32_t ld, int32_t s_txt,
                                        float eps, void* stream) {
  DRAG_CHECK(qkv, "drag_k_norm_rope_vt_bf16: null pointer");      // vt == null: k only (drag_attention_v_bf16 reads v as it is)
  DRAG_CHECK((wk_txt == nullptr) == (wk_img == nullptr), "drag_k_norm_rope_vt_bf16: norm weights come in pairs");
  DRAG_CHECK((rope_cos == nullptr) == (rope_sin == nullptr), "drag_k_norm_rope_vt_bf16: cos/sin come in pairs");
  DRAG_CHECK(B > 0 && S > 0 && H > 0 && ld >= 3 * H * 128 && ld % 8 == 0, "drag_k_norm_rope_vt_bf16: bad shape");
  PrepArgs p;
  p.qkv = (bf16_t*)qkv; p.vt = (bf16_t*)vt;
  p.wq_txt = (const bf16_t*)wk_txt; p.wk_txt = (const bf16_t*)wk_txt;      // wq_* only flags "normalise" in the kernel
  p.wq_img = (const bf16_t*)wk_img; p.wk_img = (const bf16_t*)wk_img;
  p.cosT = rope_cos; p.sinT = rope_sin;
  p.B = B; p.S = S; p.H = H; p.ld = ld; p.s_txt = s_txt; p.s_pad = (S + 63) / 64 * 64; p.eps = eps;
  p.skip_q = 1;
  hipLaunchKernelGGL(qk_norm_rope_vt_kernel, dim3((S + 63) / 64, H, B), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}

static int attention_launch(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t S, int32_t H,
                            int32_t ld_qk, int64_t qk_batch_stride, int32_t ld_o, int64_t o_batch_stride, float scale,
                            const void* wq_txt, const void* wq_img, const float* rope_cos, const float* rope_sin, int32_t s_txt,
                            float eps, void* stream, bool vrow = false);

// which kernel family a call of S keys takes (one policy for the launch and for drag_attention_bf16_choice)
static void attention_family(int32_t S, bool vrow, bool& w8, bool& q64) {
  w8 = !drag_opt(DRAG_OPT_ATTN_W4) && S >= 4096;     // 8-wave blocks halve the DMA issue per wave; below ~4k keys 128-query blocks balance better (S=1753: 945 vs 843 TFLOP/s)
  // "attn_q64": 0 = policy (the 4-wave x 64-query kernel for S >= 4096, where it is 4-5 % ahead; below that its 256-query blocks fill
  // the chip worse: S = 1753 911 vs 982 TFLOP/s), 1 = whenever S >= 1024, 2 = never (the 8-wave / 4-wave x 32-query family: measurements, tests)
  const int q64opt = drag_opt(DRAG_OPT_ATTN_Q64);
  q64 = !vrow && q64opt != 2 && ((q64opt == 1 && S >= 1024) || (q64opt == 0 && w8 && (!DRAG_EXP || drag_opt(DRAG_OPT_ATTN_PERSIST) == 0)));
}

// does a 64-query launch over S keys take the generated stream (attention_q64g_kernel)?  Its last two tiles stage the next item's first
// tiles: an even number >= 4 of KV tiles; "attn_gen" = 1 switches it off
static bool attention_generated(int32_t S) {
  const int nkv = (S + 63) / 64;
  return drag_opt(DRAG_OPT_ATTN_GEN) != 1 && nkv % 2 == 0 && nkv >= 4;
}

// 64 = attention_q64_kernel, 640 = attention_q64g_kernel without the fold, 641 = with it (fused q preparation only), 8 / 4 =
// attention_d128_kernel<8 | 4 waves, ...> — for callers that account launches per kernel (bench.py)
extern "C" int drag_attention_bf16_choice(int32_t S, int32_t v_row_major, int32_t q_prep) {
  bool w8, q64;
  attention_family(S, v_row_major != 0, w8, q64);
  if (q64 && attention_generated(S)) return q_prep && drag_opt(DRAG_OPT_ATTN_GEN) != 2 ? 641 : 640;
  return q64 ? 64 : (w8 ? 8 : 4);
}

extern "C" int drag_attention_bf16(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t S,
                                   int32_t H, int32_t ld_qk, int64_t qk_batch_stride, int32_t ld_o,
                                   int64_t o_batch_stride, float scale, void* stream) {
  return attention_launch(q, k, vt, out, B, S, H, ld_qk, qk_batch_stride, ld_o, o_batch_stride, scale, nullptr, nullptr, nullptr,
                          nullptr, 0, 0.f, stream);
}

// attention whose Q operand is the RAW q projection: per-head RMSNorm (weights wq_txt for rows < s_txt, wq_img after) and
// RoPE are applied while the Q fragments are loaded (FluxAttnProcessor2_0: norm_q / norm_added_q + apply_rotary_emb)
extern "C" int drag_attention_qprep_bf16(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t S,
                                         int32_t H, int32_t ld_qk, int64_t qk_batch_stride, int32_t ld_o,
                                         int64_t o_batch_stride, float scale, const void* wq_txt, const void* wq_img,
                                         const float* rope_cos, const float* rope_sin, int32_t s_txt, float eps, void* stream) {
  DRAG_CHECK(wq_txt && wq_img && rope_cos && rope_sin,
             "drag_attention_qprep_bf16: the fused q preparation needs both norm weights and both RoPE tables "
             "(drag_attention_bf16 takes an already prepared q)");
  DRAG_CHECK(s_txt >= 0 && s_txt <= S, "drag_attention_qprep_bf16: 0 <= s_txt <= S");
  return attention_launch(q, k, vt, out, B, S, H, ld_qk, qk_batch_stride, ld_o, o_batch_stride, scale, wq_txt, wq_img, rope_cos,
                          rope_sin, s_txt, eps, stream);
}

// attention over q | k | v AS THE LINEARS WROTE THEM: v is read row-major from the projection buffer (same row stride and batch
// stride as q / k) — no V^T pass, no V^T buffer.  The q preparation is optional: all four of wq_txt / wq_img / rope_cos /
// rope_sin, or none of them (then q is used as stored, like drag_attention_bf16).
extern "C" int drag_attention_v_bf16(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t S, int32_t H,
                                     int32_t ld_qk, int64_t qk_batch_stride, int32_t ld_o, int64_t o_batch_stride, float scale,
                                     const void* wq_txt, const void* wq_img, const float* rope_cos, const float* rope_sin,
                                     int32_t s_txt, float eps, void* stream) {
  const int given = (wq_txt != nullptr) + (wq_img != nullptr) + (rope_cos != nullptr) + (rope_sin != nullptr);
  DRAG_CHECK(given == 0 || given == 4, "drag_attention_v_bf16: the fused q preparation takes both norm weights and both RoPE tables, or none");
  DRAG_CHECK(given == 0 || (s_txt >= 0 && s_txt <= S), "drag_attention_v_bf16: 0 <= s_txt <= S");
  return attention_launch(q, k, v, out, B, S, H, ld_qk, qk_batch_stride, ld_o, o_batch_stride, scale, wq_txt, wq_img, rope_cos,
                          rope_sin, s_txt, eps, stream, true);
}

#if DRAG_EXP
// workgroups per XCD of the persistent attention kernel
static int persist_slots() {
  const int opt = drag_opt(DRAG_OPT_ATTN_PERSIST);
  if (opt >= 3) return opt;           // (1 = one per CU)
  static int ncu8 = 0;
  if (ncu8 == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    ncu8 = n / 8 > 0 ? n / 8 : 1;
  }
  return ncu8;
}
#endif

static int attention_launch(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t S, int32_t H,
                            int32_t ld_qk, int64_t qk_batch_stride, int32_t ld_o, int64_t o_batch_stride, float scale,
                            const void* wq_txt, const void* wq_img, const float* rope_cos, const float* rope_sin, int32_t s_txt,
                            float eps, void* stream, bool vrow) {
  DRAG_CHECK(q && k && vt && out, "drag_attention_bf16: null pointer");
  DRAG_CHECK(B > 0 && S > 0 && H > 0, "drag_attention_bf16: bad shape");
  DRAG_CHECK(ld_qk % 8 == 0 && ld_o % 4 == 0, "drag_attention_bf16: ld_qk %% 8, ld_o %% 4 required");
  AttnArgs p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.v = (const bf16_t*)vt; p.out = (bf16_t*)out;
  p.B = B; p.S = S; p.H = H; p.ld_qk = ld_qk; p.ld_o = ld_o; p.s_pad = (S + 63) / 64 * 64;
  p.qk_bs = qk_batch_stride; p.o_bs = o_batch_stride;
  p.c = scale * 1.4426950408889634f;
  p.wq_txt = (const bf16_t*)wq_txt; p.wq_img = (const bf16_t*)wq_img; p.cosT = rope_cos; p.sinT = rope_sin;
  p.s_txt = s_txt; p.eps = eps;
  const long long kspan = ((long long)(S - 1) * ld_qk + 128) * 2;
  const long long vspan = (long long)128 * p.s_pad * 2;
  DRAG_CHECK(kspan < (1ll << 31), "drag_attention_bf16: K span must be < 2 GiB per (batch, head)");
  p.k_bytes = (unsigned)kspan; p.vt_bytes = (unsigned)vspan;
  const long long kall = ((long long)(B - 1) * qk_batch_stride + (H - 1) * 128) * 2 + kspan, vall = (long long)B * H * vspan;
  // (whole-tensor spans: only the persistent experiment addresses through them — its launch condition below checks they fit 32 bits)
  const bool fits32 = kall < (1ll << 32) && vall < (1ll << 32);
  p.k_bytes_all = fits32 ? (unsigned)kall : 0u; p.vt_bytes_all = fits32 ? (unsigned)vall : 0u;
  const int groups = (B * H + 7) / 8;
  bool w8, q64;
  attention_family(S, vrow, w8, q64);
  const int QB = (w8 || q64) ? 256 : 128;
  const int nqb2 = (S + QB - 1) / QB;
  const dim3 grid(8 * groups * nqb2);
  const int sched = drag_opt(DRAG_OPT_ATTN_SCHED);          // 0, 1, or 2 = schedule 1 + pipelined row maxima
  const bool qprep = wq_txt != nullptr;
  p.tune = drag_opt(DRAG_OPT_ATTN_TUNE);
  if (ld_o % 8 != 0 || ((uintptr_t)out & 15) != 0 || o_batch_stride % 8 != 0) p.tune &= ~2;     // 16-byte stores need 16-byte aligned rows
  const hipStream_t st = (hipStream_t)stream;
#define DRAG_ATTN_LAUNCH(NW, SC, QP, PM) hipLaunchKernelGGL((attention_d128_kernel<NW, SC, QP, PM>), grid, dim3(NW * 64), 0, st, p)
#define DRAG_ATTN_PICK(NW)                                                                           \
  do {                                                                                               \
    if (DRAG_EXP && sched == 3) { if (qprep) DRAG_ATTN_LAUNCH(NW, DRAG_EXP ? 2 : 1, true, true); else DRAG_ATTN_LAUNCH(NW, DRAG_EXP ? 2 : 1, false, true); }        \
    else if (sched >= 2) { if (qprep) DRAG_ATTN_LAUNCH(NW, 1, true, true); else DRAG_ATTN_LAUNCH(NW, 1, false, true); }   \
    else if (sched == 1) { if (qprep) DRAG_ATTN_LAUNCH(NW, 1, true, false); else DRAG_ATTN_LAUNCH(NW, 1, false, false); } \
    else { if (qprep) DRAG_ATTN_LAUNCH(NW, 0, true, false); else DRAG_ATTN_LAUNCH(NW, 0, false, false); }                 \
  } while (0)
  if (vrow) {                   // row-major V: the product schedule only (SCHED 1 + PMAX)
    if (w8) { if (qprep) hipLaunchKernelGGL((attention_d128_kernel<8, 1, true, true, true>), grid, dim3(512), 0, st, p);
              else hipLaunchKernelGGL((attention_d128_kernel<8, 1, false, true, true>), grid, dim3(512), 0, st, p); }
    else { if (qprep) hipLaunchKernelGGL((attention_d128_kernel<4, 1, true, true, true>), grid, dim3(256), 0, st, p);
           else hipLaunchKernelGGL((attention_d128_kernel<4, 1, false, true, true>), grid, dim3(256), 0, st, p); }
  } else if (q64) {
    // 64 KiB of tiles + 64 KiB for the q rows of the next item (LDS-DMA).  One workgroup per CU walks the items (persistent) when every
    // item exists (B * H a multiple of 8) and the tiles pair up (an even number, >= 4: the stream's last two tiles stage for the next item);
    // otherwise, and under "attn_walk" = 2, one item per workgroup
    constexpr int lds64 = 2 * KT_BYTES + 2 * VT_BYTES + 4 * 16384;
    static unsigned long long ready64 = 0;       // one bit per device: the attribute belongs to the device's copy of the kernel
    static int ncu64 = 0;
    int dev64 = 0;
    (void)hipGetDevice(&dev64);
    if (!((ready64 >> (dev64 & 63)) & 1ull)) {
      DRAG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_q64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64) == hipSuccess &&
                 hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_q64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64) == hipSuccess,
                 "drag_attention: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      ready64 |= 1ull << (dev64 & 63);
    }
    if (ncu64 == 0) {
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev64) != hipSuccess || n <= 0) n = 256;
      ncu64 = (n & ~7) ? (n & ~7) : 8;
    }
    p.items = (int)grid.x;
    const int nkv64 = p.s_pad / 64;
    const int popt = drag_opt(DRAG_OPT_ATTN_WALK);             // 0 policy | 2 one item per workgroup | n >= 8 (a multiple of 8): n workgroups (tests: many items each)
    const int pgrid = popt >= 8 && popt % 8 == 0 ? popt : ncu64;
    // (a forced grid also walks items whose tiles do not pair up: each item then stages its own first tiles behind a barrier — tests)
    const bool walk = popt != 2 && (B * H) % 8 == 0 && (popt >= 8 || (nkv64 % 2 == 0 && nkv64 >= 4)) && p.items > pgrid;
    const dim3 grid64(walk ? (unsigned)pgrid : grid.x);
    // "attn_gen": 0 = policy — the generated stream (attention_q64g_kernel) whenever the tiles pair up, FOLD with the fused q preparation;
    // 1 = never (attention_q64_kernel); 2 = the generated stream WITHOUT the fold (same bits as attention_q64_kernel: tests, A/B)
    const int gen = drag_opt(DRAG_OPT_ATTN_GEN);
    if (attention_generated(S)) {
      static unsigned long long readyg = 0;
      if (!((readyg >> (dev64 & 63)) & 1ull)) {
        DRAG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_q64g_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_q64g_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_q64g_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64) == hipSuccess,
                   "drag_attention: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        readyg |= 1ull << (dev64 & 63);
      }
#if DRAG_EXP
      if (qprep && gen >= 11 && gen <= 29) {
#define DRAG_AQV(V_) case 10 + V_: DRAG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_q64g_kernel<true, true, V_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64) == hipSuccess, "hipFuncSetAttribute"); \
        hipLaunchKernelGGL((attention_q64g_kernel<true, true, V_>), grid64, dim3(256), lds64, st, p); break
        switch (gen) { DRAG_AQV(1); DRAG_AQV(2); DRAG_AQV(3); DRAG_AQV(4); DRAG_AQV(5); DRAG_AQV(6); DRAG_AQV(7); DRAG_AQV(8); DRAG_AQV(9); DRAG_AQV(10); DRAG_AQV(11); DRAG_AQV(12); DRAG_AQV(13); DRAG_AQV(14); DRAG_AQV(15); DRAG_AQV(16); DRAG_AQV(17); DRAG_AQV(18); DRAG_AQV(19); }
#undef DRAG_AQV
      } else
#endif
      if (qprep && gen != 2) hipLaunchKernelGGL((attention_q64g_kernel<true, true>), grid64, dim3(256), lds64, st, p);
      else if (qprep) hipLaunchKernelGGL((attention_q64g_kernel<true, false>), grid64, dim3(256), lds64, st, p);
      else hipLaunchKernelGGL((attention_q64g_kernel<false, false>), grid64, dim3(256), lds64, st, p);
    } else if (qprep) hipLaunchKernelGGL((attention_q64_kernel<true>), grid64, dim3(256), lds64, st, p);
    else hipLaunchKernelGGL((attention_q64_kernel<false>), grid64, dim3(256), lds64, st, p);
#if DRAG_EXP
  } else if (w8 && sched == 2 && drag_opt(DRAG_OPT_ATTN_PERSIST) != 0 && fits32 && (B * H) % 8 == 0 && (p.s_pad / 64) % 2 == 0 &&
             nqb2 * groups > persist_slots()) {
    // EXPERIMENT, off by default ("attn_persist": 0 = off, 1 = one workgroup per CU, n >= 3 = n workgroups per XCD — tests: many items
    // per workgroup): the persistent form, whose K / V^T stream runs on across a workgroup's items (the KV loop is unrolled by two: an
    // even number of tiles keeps the buffer parity across items).  Same bits either way.  Measured same-box, interleaved
    // (scripts/ab_attn_persist.py, B=8 S=5337): 2592 vs 2571 us with the q preparation, 2523 vs 2499 us without — 1 % SLOWER: what a
    // workgroup pays outside its KV loop (6 us plain, 11-12 us with the q preparation) is the round trip + ingest of its Q rows and
    // RoPE table rows (256 KiB per workgroup through one CU's ~100 GB/s global-load path), not the first K / V^T tiles nor the
    // dispatch; and the hardware already overlaps the seams of one-item workgroups (a CU holds two of them by LDS and registers, so
    // the next workgroup's waves start on SIMDs as the old one's retire), which a single persistent workgroup cannot do.
    const dim3 pgrid(8 * persist_slots());
    if (qprep) hipLaunchKernelGGL((attention_d128_kernel<8, 1, true, true, false, true>), pgrid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((attention_d128_kernel<8, 1, false, true, false, true>), pgrid, dim3(512), 0, st, p);
#endif  // DRAG_EXP
  } else if (w8) DRAG_ATTN_PICK(8); else DRAG_ATTN_PICK(4);
#undef DRAG_ATTN_PICK
#undef DRAG_ATTN_LAUNCH
  DRAG_LAUNCH_CHECK();
  return 0;
}
