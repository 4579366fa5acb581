// gemm_bf16.hip — C[M,N] = epilogue(A[M,K] · W[N,K]^T) on gfx950 MFMA (bf16 in, fp32 accumulate).
//
// Replaces every torch.nn.Linear the reference reaches through diffusers/transformers/clip
// (FluxTransformerBlock / FluxSingleTransformerBlock projections and MLPs,
// embedders, SigLIP / CLIP ViT linears; call sites batch_generate_flux_kshot.py:467-474 and
// outpainting_updown_sampling_redux.py:1246-1257).  Both operands are K-contiguous
// (torch Linear weight layout), which is the natural MFMA operand layout.
//
// Structure (kernel "t128"): 256 threads = 4 waves (2x2), 128x128x64 tile, each wave a 64x64
// sub-tile as 4x4 v_mfma_f32_16x16x32_bf16.  Operand tiles go HBM -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), double-buffered, one barrier
// per K-tile.  The LDS image is lane-linear, so the bank-conflict XOR swizzle is applied to
// the per-lane *source* address and again on the ds_read_b128 (involution).
// MFMA operands are swapped (a = W fragment, b = A fragment) so each lane ends up holding 4
// consecutive output columns -> 8-byte bf16 stores and vector bias/gate/residual loads.
//
// Row addressing is "batched rows": logical row r lives at base + (r / rpb) * bs + (r % rpb) * ld,
// which lets the text and image streams of Flux live inside one joint [B, S, D] buffer with no
// concat copies.
#include "drag_common.h"
// The asm statements that write m0 (one s_add_u32 m0 per LDS-DMA piece) list "m0" as a clobber: hipcc then re-materialises m0 before its own
// next LDS-DMA builtin (checked on a two-builtin probe: without the clobber the second builtin ran on the asm's stale m0).  clang warns that m0
// is a reserved register on every such statement; the clobber is what is wanted here.
#pragma clang diagnostic ignored "-Winline-asm"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct RowMap {
  int rpb;        // rows per batch
  long long bs;   // batch stride (elements)
  int ld;         // row stride (elements)
  __device__ __forceinline__ long long off(int r) const {
    int b = r / rpb;
    int s = r - b * rpb;
    return (long long)b * bs + (long long)s * ld;
  }
};

// conv mode (implicit GEMM, 3x3): A is a zero-haloed NHWC activation [B, Hp, Wp, Cin]; logical row
// m = (b, y, x) of the [B*Ho*Wo, 9*Cin] im2col matrix starts at pixel (y*stride + oy, x*stride + ox)
// and K-tile kt = (tap, channel chunk) adds ((tap/3)*Wp + tap%3)*Cin + chunk*64.
struct ConvMap {
  int Ho, Wo, Hp, Wp, Cin, stride, oy, ox;
  __device__ __forceinline__ long long off(int m) const {
    const int hw = Ho * Wo;
    const int b = m / hw;
    const int r = m - b * hw;
    const int y = r / Wo;
    const int x = r - y * Wo;
    return (((long long)b * Hp + y * stride + oy) * Wp + x * stride + ox) * Cin;
  }
};

struct GemmKArgs {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const bf16_t* bias;   // [N] or null
  const bf16_t* gate;   // [batch, ldg] or null:  C = resid + gate[b, n] * (acc + bias)
  const bf16_t* resid;  // same row addressing as C, or null
  int M, N, K;
  RowMap am, cm;
  ConvMap cv;
  int ldg;
  int act;
  int act_n0;     // activation applies to columns >= act_n0
  int out_f32;
  unsigned a_bytes, w_bytes;
  int tiles_m, tiles_n;
  int wide;       // C / resid / gate rows are 16-B aligned and N % 8 == 0: staged epilogue
  // optional second destination: output columns >= n_split go to C2 (dense rows of ld2 elements, column n -> C2[n - n_split]).
  // Lets two Linears over the same input run as ONE launch into two buffers (Flux single blocks: to_q|k|v and proj_mlp).
  void* C2;
  int ld2, n_split;
  int group_m;    // M tiles per group of the tile walk (8; "gemm_group_m" option for measurements)
  int ldw;          // W's row stride in elements (= K, except in a split-K launch: the whole K of the Linear)
  long long w_boff; // split-K launch (gemm_bf16_w4p only): what row batch b of A adds to W's base (elements): batch b multiplies columns b K .. b K + K - 1
  int split_m1;     // split-K launch of a PAIR: rows >= split_m1 of every K slice are the second problem's (operands A2 / W2, same row stride); 0: one problem
  int w4_late_state;   // "gemm_epilogue" = 2 (measurement): gemm_bf16_w4p computes tile t + 2's state between tile t's K loop and its epilogue instead of in front of tile t + 1's K loop
  int epi_generic; // "gemm_epilogue" = 1: every tile takes the general staged epilogue (tests compare it with the specialised one bit for bit)
  // optional second row segment (drag_gemm_bf16_pair): M tiles >= seg_tiles_m belong to a second problem with its own operands and
  // row maps but the same N, K and epilogue form — a double block's text and image Linears as ONE launch of the non-persistent kernels
  int seg_tiles_m;          // 0: one segment
  const bf16_t* A2;
  const bf16_t* W2;
  void* Cs2;
  const bf16_t* bias2;
  const bf16_t* gate2;
  const bf16_t* resid2;
  int M2, ldg2, wide2;
  RowMap am2, cm2;
};

// a workgroup whose M tile lies in the second segment swaps that segment's operands in (wave-uniform: scalar moves)
__device__ __forceinline__ void pick_segment(GemmKArgs& p, int& tm) {
  if (p.seg_tiles_m > 0 && tm >= p.seg_tiles_m) {
    tm -= p.seg_tiles_m;
    p.A = p.A2; p.W = p.W2; p.C = p.Cs2; p.bias = p.bias2; p.gate = p.gate2; p.resid = p.resid2;
    p.M = p.M2; p.ldg = p.ldg2; p.wide = p.wide2; p.am = p.am2; p.cm = p.cm2;
  }
}

// the arguments as the epilogue of the tile at column n0 sees them
__device__ __forceinline__ GemmKArgs dest_of(const GemmKArgs& p, int n0) {
  GemmKArgs q = p;
  if (p.C2 != nullptr && n0 >= p.n_split) {
    q.C = (void*)((bf16_t*)p.C2 - p.n_split);
    q.cm.ld = p.ld2;
  }
  return q;
}


// per-column epilogue operands of one 4-wide column group, loaded once per tile column (not once per row)
struct ColOps {
  float b[4];      // bias
  float g[4];      // gate (valid when the tile lies inside one batch)
};
__device__ __forceinline__ void load_colops(const GemmKArgs& p, int n, int bidx, bool gate_uniform, ColOps& c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) { c.b[r] = 0.f; c.g[r] = 0.f; }
  if (n >= p.N) return;
  if (p.bias) {
    const u32x2_t bb = *(const u32x2_t*)(p.bias + n);
    c.b[0] = bf2f((bf16_t)(bb[0] & 0xffff)); c.b[1] = bf2f((bf16_t)(bb[0] >> 16));
    c.b[2] = bf2f((bf16_t)(bb[1] & 0xffff)); c.b[3] = bf2f((bf16_t)(bb[1] >> 16));
  }
  if (p.gate && gate_uniform) {
    const u32x2_t gg = *(const u32x2_t*)(p.gate + (long long)bidx * p.ldg + n);
    c.g[0] = bf2f((bf16_t)(gg[0] & 0xffff)); c.g[1] = bf2f((bf16_t)(gg[0] >> 16));
    c.g[2] = bf2f((bf16_t)(gg[1] & 0xffff)); c.g[3] = bf2f((bf16_t)(gg[1] >> 16));
  }
}

// The fused activations are all  y = x * sigmoid(x * (c0 + c1 x^2)):  GELU-tanh (c0, c1) = (2k, 2k*0.044715),
// SiLU (1, 0), QuickGELU (1.702, 0) -> one branch-free body, tiny code (the epilogue is inlined 32x per lane;
// a switch over libm-style bodies there blew the instruction cache and cost >25 % on K = 3072 GEMMs).
struct ActCoef { float c0, c1; };
__device__ __forceinline__ ActCoef act_coef(int act) {
  switch (act) {
    case DRAG_ACT_GELU_TANH: return {2.0f * 0.7978845608028654f, 2.0f * 0.7978845608028654f * 0.044715f};
    case DRAG_ACT_SILU: return {1.0f, 0.0f};
    case DRAG_ACT_QUICK_GELU: return {1.702f, 0.0f};
    default: return {0.0f, 0.0f};
  }
}

// The activation of four consecutive columns, two values per instruction: the operations of  x * fast_sigmoid(x * (c0 + c1 x x))  in the
// scalar order — (c1 x), fma(.., x, c0), x *, * (-log2 e), 2^, 1 +, 1 /, x * — so the bits are those of the scalar body; the six
// non-transcendental ones become v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 (the compiler's own vectoriser stops at the v_exp / v_rcp
// pair: 7.5 instructions per value, 5 here; the GELU epilogue of a 256 x 256 tile was 3200 instructions per wave, a tenth of the K = 3072 tile)
__device__ __forceinline__ void act4(float* v, ActCoef ac) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // torch: y = linear(x) is a bf16 tensor before the activation reads it
    const f32x2_t x = {rbf(v[2 * h]), rbf(v[2 * h + 1])};
    const f32x2_t m1 = ac.c1 * x;
    const f32x2_t t = __builtin_elementwise_fma(m1, x, (f32x2_t){ac.c0, ac.c0});
    const f32x2_t z = x * t;
    const f32x2_t a = -1.4426950408889634f * z;
    const f32x2_t e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const f32x2_t d = 1.0f + e;
    const f32x2_t r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2_t y = x * r;
    v[2 * h] = y[0];
    v[2 * h + 1] = y[1];
  }
}

// epilogue of one accumulator row-group: NI groups of 4 consecutive columns of ONE output row.
// CHECK = false is the interior-tile fast path (no bounds tests, residual loads issued up front).
template <int NI, bool CHECK>
__device__ __forceinline__ void epi_row(const GemmKArgs& p, long long coff, int bidx, int nbase, const f32x4_t* a,
                                        const ColOps* c, bool gate_uniform, ActCoef ac) {
  u32x2_t rr[NI];
  if (p.resid) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = nbase + ni * 16;
      rr[ni] = (u32x2_t){0u, 0u};
      if (!CHECK || n < p.N) rr[ni] = *(const u32x2_t*)(p.resid + coff + n);
    }
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = nbase + ni * 16;
    if (CHECK && n >= p.N) continue;
    float v[4] = {a[ni][0] + c[ni].b[0], a[ni][1] + c[ni].b[1], a[ni][2] + c[ni].b[2], a[ni][3] + c[ni].b[3]};
    if (p.act != DRAG_ACT_NONE && n >= p.act_n0) {
      act4(v, ac);
    }
    if (p.gate) {
      // diffusers computes  x = x + gate * y  with y, gate, x bf16 tensors: y is rounded to
      // bf16 first, the product is rounded, then the sum is rounded.
      float g[4] = {c[ni].g[0], c[ni].g[1], c[ni].g[2], c[ni].g[3]};
      if (!gate_uniform) {
        const u32x2_t gg = *(const u32x2_t*)(p.gate + (long long)bidx * p.ldg + n);
        g[0] = bf2f((bf16_t)(gg[0] & 0xffff)); g[1] = bf2f((bf16_t)(gg[0] >> 16));
        g[2] = bf2f((bf16_t)(gg[1] & 0xffff)); g[3] = bf2f((bf16_t)(gg[1] >> 16));
      }
      const float x[4] = {bf2f((bf16_t)(rr[ni][0] & 0xffff)), bf2f((bf16_t)(rr[ni][0] >> 16)),
                          bf2f((bf16_t)(rr[ni][1] & 0xffff)), bf2f((bf16_t)(rr[ni][1] >> 16))};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = x[r] + rbf(g[r] * rbf(v[r]));
    } else if (p.resid) {
      v[0] = bf2f((bf16_t)(rr[ni][0] & 0xffff)) + rbf(v[0]); v[1] = bf2f((bf16_t)(rr[ni][0] >> 16)) + rbf(v[1]);
      v[2] = bf2f((bf16_t)(rr[ni][1] & 0xffff)) + rbf(v[2]); v[3] = bf2f((bf16_t)(rr[ni][1] >> 16)) + rbf(v[3]);
    }
    if (p.out_f32) {
      *(f32x4_t*)((float*)p.C + coff + n) = (f32x4_t){v[0], v[1], v[2], v[3]};
    } else {
      u32x2_t o;
      o[0] = pack2bf(v[0], v[1]);
      o[1] = pack2bf(v[2], v[3]);
      *(u32x2_t*)((bf16_t*)p.C + coff + n) = o;
    }
  }
}

// whole-wave epilogue: MI row groups x NI column groups; rows m = mrow0 + 16*mi, columns nbase + 16*ni
template <int MI, int TM, int TN = TM, int NI = 4>
__device__ __forceinline__ void wave_epilogue(const GemmKArgs& p, int m0, int mrow0, int n0, int nbase, f32x4_t (*acc)[NI]) {
  const int b_first = m0 / p.cm.rpb;
  const bool gate_uniform = b_first == (min(m0 + TM, p.M) - 1) / p.cm.rpb;     // whole tile inside one batch
  const ActCoef ac = act_coef(p.act);
  ColOps co[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) load_colops(p, nbase + ni * 16, b_first, gate_uniform, co[ni]);
  const bool interior = m0 + TM <= p.M && n0 + TN <= p.N;
  if (interior) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = mrow0 + mi * 16;
      epi_row<NI, false>(p, p.cm.off(m), m / p.cm.rpb, nbase, acc[mi], co, gate_uniform, ac);
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = mrow0 + mi * 16;
      if (m >= p.M) continue;
      epi_row<NI, true>(p, p.cm.off(m), m / p.cm.rpb, nbase, acc[mi], co, gate_uniform, ac);
    }
  }
}

// ---- staged epilogue (bf16 output, 16-B aligned rows) -------------------------------------------------------------
// Stores are priced per cache line touched per instruction (measured: the fragment-layout epilogue above, 16 rows x
// 32 B per store instruction, cost 9.7 us of a 256x256 tile's ~75 us at K = 3072 — 58 us of a 460 us GEMM — and the
// same with every tile aimed at one L2-resident location, i.e. issue-bound, not HBM-bound).  So each wave transposes
// its 128x64 sub-tile through a private 2 KiB LDS slab, 16 rows (one MFMA row block) at a time:
//   fragment side: v = acc + bias, activation, round to bf16 (every consumer below reads the bf16 value, as torch's
//                  bf16 linear output), ds_write_b64 of 4 columns;
//   row side     : lane (row l>>3 (+8), 16-B chunk l&7) reads 8 consecutive columns back, applies gate / residual
//                  with 16-B loads and stores 16 B: one store instruction = 8 full 128-B lines.
// Slab layout: row r at r*128 B; its 16-B slots are XOR-swizzled with (r & 7) and the 8-B halves of a slot with
// (r >> 3), which makes the 16-lane ds_write_b64 groups and the ds_read_b128 groups bank-conflict free.
__device__ __forceinline__ float bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// A wave with more than 4 column blocks (the 192-column tiles of gemm_bf16_deep: NI = 6) runs the slab pass per GROUP of <= 4
// blocks: group (NI0, NIG) covers the wave's columns 16 * NI0 .. 16 * (NI0 + NIG); nw0 is the group's first column.  A group of
// 2 blocks uses the same slab layout with the upper half of its columns (and of the row-side lanes) idle.
template <int MI, int TM, bool CHECK, int NI = 4, int NI0 = 0, int NIG = 4>
__device__ __forceinline__ void staged_rows(const GemmKArgs& p, int m0, int mw0, int n0, int nw0, int l, f32x4_t (*acc)[NI],
                                            char* scr) {
  const int q = l >> 4, r16 = l & 15;
  const int c = l & 7, rl = l >> 3;
  const ActCoef ac = act_coef(p.act);
  const int b_first = m0 / p.cm.rpb;
  const bool one_batch = b_first == (min(m0 + TM, p.M) - 1) / p.cm.rpb;     // whole tile inside one batch
  // fragment side: bias of this lane's 4 columns per column block
  float bias[4][4];
  bool actv[4];
#pragma unroll
  for (int ni = 0; ni < NIG; ++ni) {
    const int n = nw0 + ni * 16 + q * 4;
    bias[ni][0] = bias[ni][1] = bias[ni][2] = bias[ni][3] = 0.f;
    if (p.bias && (!CHECK || n < p.N)) {
      const u32x2_t bb = *(const u32x2_t*)(p.bias + n);
      bias[ni][0] = bf_lo(bb[0]); bias[ni][1] = bf_hi(bb[0]); bias[ni][2] = bf_lo(bb[1]); bias[ni][3] = bf_hi(bb[1]);
    }
    actv[ni] = p.act != DRAG_ACT_NONE && n >= p.act_n0;
  }
  int woff[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
    woff[ni] = r16 * 128 + (((2 * ni + (q >> 1)) ^ (r16 & 7)) << 4) + (((q & 1) ^ (r16 >> 3)) << 3);
  const int roff = rl * 128 + ((c ^ rl) << 4);                // + j * 1024; halves swapped for j = 1
  // row side: this lane's 8 columns
  const int n = nw0 + c * 8;
  const bool col_ok = (!CHECK || n + 8 <= p.N) && (NIG == 4 || c * 8 < NIG * 16);   // N % 8 == 0 on this path; a short group's upper lanes idle
  float g[8];
  if (p.gate && one_batch && col_ok) {
    const u32x4_t gg = *(const u32x4_t*)(p.gate + (long long)b_first * p.ldg + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[2 * i] = bf_lo(gg[i]); g[2 * i + 1] = bf_hi(gg[i]); }
  }
  const long long off0 = p.cm.off(min(mw0, p.M - 1));
  // residual rows are requested RD row blocks (passes) ahead of their use into a small register ring: one exposed HBM
  // latency per tile instead of one per pass (the per-pass form cost a gated GEMM 18 % at K = 3072); the whole tile at
  // once (64 VGPRs at MI = 8) pushed the 256x256 kernel into scratch spills inside its main loop
  constexpr int RD = MI < 2 ? MI : 2;
  u32x4_t rres[RD][2];
  auto load_resid = [&](int mi2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = mw0 + mi2 * 16 + j * 8 + rl;
      rres[mi2 % RD][j] = (u32x4_t){0u, 0u, 0u, 0u};
      if ((CHECK || NIG < 4) && ((CHECK && m >= p.M) || !col_ok)) continue;
      const long long coff = (one_batch ? off0 + (long long)(m - mw0) * p.cm.ld : p.cm.off(m)) + n;
      rres[mi2 % RD][j] = *(const u32x4_t*)(p.resid + coff);
    }
  };
  if (p.resid) {
#pragma unroll
    for (int mi = 0; mi < RD; ++mi) load_resid(mi);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NIG; ++ni) {
      float v[4] = {acc[mi][NI0 + ni][0] + bias[ni][0], acc[mi][NI0 + ni][1] + bias[ni][1], acc[mi][NI0 + ni][2] + bias[ni][2],
                    acc[mi][NI0 + ni][3] + bias[ni][3]};
      if (actv[ni]) {
        act4(v, ac);
      }
      u32x2_t o;
      o[0] = pack2bf(v[0], v[1]);
      o[1] = pack2bf(v[2], v[3]);
      *(u32x2_t*)(scr + woff[ni]) = o;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = mw0 + mi * 16 + j * 8 + rl;
      u32x4_t y = *(const u32x4_t*)(scr + roff + j * 1024);
      if (j == 1) y = (u32x4_t){y[2], y[3], y[0], y[1]};
      if ((CHECK || NIG < 4) && ((CHECK && m >= p.M) || !col_ok)) continue;
      const long long coff = (one_batch ? off0 + (long long)(m - mw0) * p.cm.ld : p.cm.off(m)) + n;
      if (p.resid) {
        const u32x4_t x = rres[mi % RD][j];
        if (p.gate) {
          // diffusers computes  x = x + gate * y  with y, gate, x bf16 tensors: the product is rounded, then the sum
          if (!one_batch) {
            const u32x4_t gg = *(const u32x4_t*)(p.gate + (long long)(m / p.cm.rpb) * p.ldg + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) { g[2 * i] = bf_lo(gg[i]); g[2 * i + 1] = bf_hi(gg[i]); }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
            y[i] = pack2bf(bf_lo(x[i]) + rbf(g[2 * i] * bf_lo(y[i])), bf_hi(x[i]) + rbf(g[2 * i + 1] * bf_hi(y[i])));
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = pack2bf(bf_lo(x[i]) + bf_lo(y[i]), bf_hi(x[i]) + bf_hi(y[i]));
        }
      }
      *(u32x4_t*)((bf16_t*)p.C + coff) = y;
    }
    if (p.resid && mi + RD < MI) load_resid(mi + RD);
  }
}

// Fast form of the staged epilogue for the common tile: interior, inside ONE batch of the output's row map, four column blocks per wave
// that are all alike (activation on all of them or on none).  Same arithmetic, operation for operation, as staged_rows — what goes is
// everything staged_rows decides at run time per row block (residual? gate? which columns are activated? does the tile cross a batch?
// the row map's integer division per store when it does): the ablations of round 4 (profiles/r04_gemm_epilogue_ablations.log) put the
// epilogue at 10 % of a K = 3072 tile with only 1.5-2.6 % of it in the LDS transpose and < 2 % in HBM writes — the rest is its own
// instruction stream.  FORM: 0 = y, 1 = resid + y, 3 = resid + gate * y;  ACT: the activation applies to every column of the tile.
template <int MI, int FORM, bool ACT, int NI = 4, int NI0 = 0>      // NI / NI0: the wave's accumulator row has NI column blocks; this call takes blocks NI0 .. NI0 + 3
__device__ __forceinline__ void staged_rows_fast(const GemmKArgs& p, long long off0, int bidx, int nw0, int l, f32x4_t (*acc)[NI], char* scr) {
  const int q = l >> 4, r16 = l & 15;
  const int c = l & 7, rl = l >> 3;
  const ActCoef ac = act_coef(p.act);
  float bias[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    bias[ni][0] = bias[ni][1] = bias[ni][2] = bias[ni][3] = 0.f;
    if (p.bias) {
      const u32x2_t bb = *(const u32x2_t*)(p.bias + nw0 + ni * 16 + q * 4);
      bias[ni][0] = bf_lo(bb[0]); bias[ni][1] = bf_hi(bb[0]); bias[ni][2] = bf_lo(bb[1]); bias[ni][3] = bf_hi(bb[1]);
    }
  }
  int woff[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
    woff[ni] = r16 * 128 + (((2 * ni + (q >> 1)) ^ (r16 & 7)) << 4) + (((q & 1) ^ (r16 >> 3)) << 3);
  const int roff = rl * 128 + ((c ^ rl) << 4);
  const int n = nw0 + c * 8;
  float g[8];
  if constexpr (FORM == 3) {
    const u32x4_t gg = *(const u32x4_t*)(p.gate + (long long)bidx * p.ldg + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[2 * i] = bf_lo(gg[i]); g[2 * i + 1] = bf_hi(gg[i]); }
  }
  // row (mi, j, rl) of the wave's sub-tile lives at off0 + (16 mi + 8 j + rl) * ld: a wave-uniform base per (mi, j) + one 32-bit lane offset
  const unsigned lane_off = (unsigned)(rl * p.cm.ld + n);
  bf16_t* const Cb = (bf16_t*)p.C + off0;
  const bf16_t* const Rb = FORM ? p.resid + off0 : nullptr;
  constexpr int RD = MI < 2 ? MI : 2;
  u32x4_t rres[RD][2];
  auto load_resid = [&](int mi2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) rres[mi2 % RD][j] = *(const u32x4_t*)(Rb + (long long)(mi2 * 16 + j * 8) * p.cm.ld + lane_off);
  };
  if constexpr (FORM != 0) {
#pragma unroll
    for (int mi = 0; mi < RD; ++mi) load_resid(mi);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float v[4] = {acc[mi][NI0 + ni][0] + bias[ni][0], acc[mi][NI0 + ni][1] + bias[ni][1], acc[mi][NI0 + ni][2] + bias[ni][2],
                    acc[mi][NI0 + ni][3] + bias[ni][3]};
      if constexpr (ACT) {
        act4(v, ac);
      }
      u32x2_t o;
      o[0] = pack2bf(v[0], v[1]);
      o[1] = pack2bf(v[2], v[3]);
      *(u32x2_t*)(scr + woff[ni]) = o;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4_t y = *(const u32x4_t*)(scr + roff + j * 1024);
      if (j == 1) y = (u32x4_t){y[2], y[3], y[0], y[1]};
      if constexpr (FORM != 0) {
        const u32x4_t x = rres[mi % RD][j];
        if constexpr (FORM == 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            y[i] = pack2bf(bf_lo(x[i]) + rbf(g[2 * i] * bf_lo(y[i])), bf_hi(x[i]) + rbf(g[2 * i + 1] * bf_hi(y[i])));
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = pack2bf(bf_lo(x[i]) + bf_lo(y[i]), bf_hi(x[i]) + bf_hi(y[i]));
        }
      }
      *(u32x4_t*)(Cb + (long long)(mi * 16 + j * 8) * p.cm.ld + lane_off) = y;
    }
    if constexpr (FORM != 0) {
      if (mi + RD < MI) load_resid(mi + RD);
    }
  }
}

// The same for a wave tile of EIGHT column blocks (the 4-wave kernel's 128 x 128): two column groups through two slabs, software-pipelined by
// hand — with one wave per SIMD nothing else covers the LDS round trip of a pass, so the slab writes of the next row block are issued
// between a group's slab reads and its stores.  Operation for operation the arithmetic of staged_rows_fast (same bits).
// EDGE: the wave's rows may end before 128 (a ragged M edge: rows >= rows_valid are neither loaded nor stored) and may cross ONE batch
// boundary of the output's row map (rows >= split belong to the next batch: base `off1 + row * ld` and the next batch's gate vector) — the
// DiT's text stream is 8 batches of 1241 rows, its joint stream 8 of 5337: tiles that straddle a batch are the rule there.  Same
// arithmetic; the interior form carries none of it.
template <int MI, int FORM, bool ACT, bool EDGE = false>
__device__ __forceinline__ void staged_rows_fast8(const GemmKArgs& p, long long off0, int bidx, int nw0, int l, f32x4_t (*acc)[8], char* scr,
                                                  int rows_valid = 1 << 30, int split = 1 << 30, long long off1 = 0) {
  const int q = l >> 4, r16 = l & 15;
  const int c = l & 7, rl = l >> 3;
  const ActCoef ac = act_coef(p.act);
  float bias[2][4][4];
#pragma unroll
  for (int grp = 0; grp < 2; ++grp)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      bias[grp][ni][0] = bias[grp][ni][1] = bias[grp][ni][2] = bias[grp][ni][3] = 0.f;
      if (p.bias) {
        const u32x2_t bb = *(const u32x2_t*)(p.bias + nw0 + 64 * grp + ni * 16 + q * 4);
        bias[grp][ni][0] = bf_lo(bb[0]); bias[grp][ni][1] = bf_hi(bb[0]); bias[grp][ni][2] = bf_lo(bb[1]); bias[grp][ni][3] = bf_hi(bb[1]);
      }
    }
  int woff[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
    woff[ni] = r16 * 128 + (((2 * ni + (q >> 1)) ^ (r16 & 7)) << 4) + (((q & 1) ^ (r16 >> 3)) << 3);
  const int roff = rl * 128 + ((c ^ rl) << 4);
  const int n = nw0 + c * 8;
  u32x4_t gq[2][2];                // gate words [batch side][group] (EDGE: both sides of the batch boundary)
  if constexpr (FORM == 3) {
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      gq[0][grp] = *(const u32x4_t*)(p.gate + (long long)bidx * p.ldg + n + 64 * grp);
      if constexpr (EDGE) gq[1][grp] = split < rows_valid && split < 128 ? *(const u32x4_t*)(p.gate + (long long)(bidx + 1) * p.ldg + n + 64 * grp) : gq[0][grp];
    }
  }
  const unsigned lane_off = (unsigned)(rl * p.cm.ld + n);
  const long long d01 = off1 - off0;         // EDGE: what a row past the boundary adds to its address (elements; >= 0: batches ascend)
  bf16_t* const Cb = (bf16_t*)p.C + off0;
  const bf16_t* const Rb = FORM ? p.resid + off0 : nullptr;
  // EDGE: the row predicate is the DESCRIPTOR's bound, not a branch (64 predicated loads / stores split the straight-line code into as
  // many basic blocks with spills between them: the first version ran as slowly as the general epilogue): accesses at or past the first
  // invalid row's byte offset are dropped / return zero by the load-store unit
  __amdgpu_buffer_rsrc_t rsC, rsR;
  unsigned d01b = 0;
  if constexpr (EDGE) {
    // (the bound is the first invalid row's FIRST byte in this wave's column range: a destination whose base is shifted — the second
    //  buffer of a two-destination launch is addressed from C2 - n_split — has valid columns beyond row_start + ld)
    const long long end = rows_valid >= 128 ? (1ll << 31) - 16 : ((long long)rows_valid * p.cm.ld + nw0 + (rows_valid > split ? d01 : 0ll)) * 2;
    rsC = __builtin_amdgcn_make_buffer_rsrc((void*)Cb, 0, (unsigned)end, 0x00020000);
    rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(FORM ? Rb : (const bf16_t*)Cb), 0, (unsigned)end, 0x00020000);
    d01b = (unsigned)(d01 * 2);
  }
  constexpr int RD = 2;
  u32x4_t rres[RD][2][2];          // [ring][group][j]
  auto load_resid = [&](int mi2) {
#pragma unroll
    for (int grp = 0; grp < 2; ++grp)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = mi2 * 16 + j * 8 + rl;
        if constexpr (EDGE) {
          const unsigned vo = (unsigned)(((mi2 * 16 + j * 8) * p.cm.ld + lane_off + 64 * grp) * 2) + (r >= split ? d01b : 0u);
          rres[mi2 % RD][grp][j] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)vo, 0, 0);
        } else {
          rres[mi2 % RD][grp][j] = *(const u32x4_t*)(Rb + (long long)(mi2 * 16 + j * 8) * p.cm.ld + lane_off + 64 * grp);
        }
      }
  };
  auto slab_values = [&](int mi, int grp, u32x2_t* o) {      // the arithmetic of one slab (registers only)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float v[4] = {acc[mi][4 * grp + ni][0] + bias[grp][ni][0], acc[mi][4 * grp + ni][1] + bias[grp][ni][1],
                    acc[mi][4 * grp + ni][2] + bias[grp][ni][2], acc[mi][4 * grp + ni][3] + bias[grp][ni][3]};
      if constexpr (ACT) {
        act4(v, ac);
      }
      o[ni][0] = pack2bf(v[0], v[1]);
      o[ni][1] = pack2bf(v[2], v[3]);
    }
  };
  auto slab_store = [&](int grp, const u32x2_t* o) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) *(u32x2_t*)(scr + grp * 2048 + woff[ni]) = o[ni];
  };
  auto write_slab = [&](int mi, int grp) {
    u32x2_t o[4];
    slab_values(mi, grp, o);
    slab_store(grp, o);
  };
  if constexpr (FORM != 0) {
#pragma unroll
    for (int mi = 0; mi < RD; ++mi) load_resid(mi);
  }
  write_slab(0, 0);
  write_slab(0, 1);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
      u32x4_t y[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        y[j] = *(const u32x4_t*)(scr + grp * 2048 + roff + j * 1024);
      }
      // The slab's two reads are ISSUED; the next row block's arithmetic runs under their latency, and its writes to the same slab follow
      // without a wait: the LDS operations of a wave execute in issue order, so a write issued behind a read cannot overtake it.  (Round 5
      // measured the form that waited for the reads first — 16 exposed LDS round trips per tile, 7300 cycles for 707 instructions.)
      __builtin_amdgcn_sched_barrier(0);
      if (mi + 1 < MI) {
        u32x2_t o[4];
        slab_values(mi + 1, grp, o);
        __builtin_amdgcn_sched_barrier(0);
        slab_store(grp, o);
      }
      __builtin_amdgcn_sched_barrier(0);
      y[1] = (u32x4_t){y[1][2], y[1][3], y[1][0], y[1][1]};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = mi * 16 + j * 8 + rl;
        if constexpr (FORM != 0) {
          const u32x4_t x = rres[mi % RD][grp][j];
          if constexpr (FORM == 3) {
            u32x4_t gw = gq[0][grp];
            if constexpr (EDGE) gw = r >= split ? gq[1][grp] : gq[0][grp];
#pragma unroll
            for (int i = 0; i < 4; ++i)
              y[j][i] = pack2bf(bf_lo(x[i]) + rbf(bf_lo(gw[i]) * bf_lo(y[j][i])), bf_hi(x[i]) + rbf(bf_hi(gw[i]) * bf_hi(y[j][i])));
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) y[j][i] = pack2bf(bf_lo(x[i]) + bf_lo(y[j][i]), bf_hi(x[i]) + bf_hi(y[j][i]));
          }
        }
        if constexpr (EDGE) {
          const unsigned vo = (unsigned)(((mi * 16 + j * 8) * p.cm.ld + lane_off + 64 * grp) * 2) + (r >= split ? d01b : 0u);
          __builtin_amdgcn_raw_buffer_store_b128(y[j], rsC, (int)vo, 0, 0);
        } else {
          *(u32x4_t*)(Cb + (long long)(mi * 16 + j * 8) * p.cm.ld + lane_off + 64 * grp) = y[j];
        }
      }
    }
    if constexpr (FORM != 0) {
      if (mi + RD < MI) load_resid(mi + RD);
    }
  }
}

template <int MI, int TM, int TN = TM, int NI = 4>      // returns true when the tile took the specialised form (a fixed number of stores per wave)
__device__ __forceinline__ bool staged_epilogue(const GemmKArgs& p, int m0, int mw0, int n0, int nw0, int l, f32x4_t (*acc)[NI],
                                                char* scr) {
  constexpr int G0 = NI < 4 ? NI : 4;
  if constexpr (NI == 4 || NI == 8) {
    const int b_first = m0 / p.cm.rpb;
    const bool fast = m0 + TM <= p.M && n0 + TN <= p.N && b_first == (m0 + TM - 1) / p.cm.rpb && (long long)8 * p.cm.ld + p.N < (1ll << 31);
    const bool act_none = p.act == DRAG_ACT_NONE || p.act_n0 >= n0 + TN, act_all = p.act != DRAG_ACT_NONE && p.act_n0 <= n0;
    if (fast && !p.epi_generic && (act_none || (act_all && !p.resid)) && !(p.gate && !p.resid)) {
      const long long off0 = p.cm.off(mw0);
      // (NI = 8, the 4-wave kernel's 128-column wave tile: two column groups of four blocks through the same slab, one after the other)
#define DRAG_FAST(FORM_, ACT_)                                                                              \
  do {                                                                                                      \
    if constexpr (NI == 8) staged_rows_fast8<MI, FORM_, ACT_>(p, off0, b_first, nw0, l, acc, scr);          \
    else staged_rows_fast<MI, FORM_, ACT_, NI, 0>(p, off0, b_first, nw0, l, acc, scr);                      \
  } while (0)
      if (!p.resid) {
        if (act_none) DRAG_FAST(0, false);
        else DRAG_FAST(0, true);
      } else if (p.gate) DRAG_FAST(3, false);
      else DRAG_FAST(1, false);
#undef DRAG_FAST
      return true;
    }
    if constexpr (NI == 8) {
      // full columns, but a ragged M edge and / or ONE batch boundary of the row map inside the tile (batches of >= 256 rows): the
      // straight-line form with a row predicate and a per-row choice between the two batches' bases / gate vectors
      const long long jump = p.cm.rpb < p.M ? p.cm.bs - (long long)p.cm.rpb * p.cm.ld : 0;      // what crossing a batch adds to a row's offset
      const bool edge = n0 + TN <= p.N && p.cm.rpb >= TM && p.cm.ld >= TN / 2 && jump >= 0 && ((long long)(TM / 2 + 8) * p.cm.ld + p.N + jump) * 2 < (1ll << 31) - 16;
      if (edge && !p.epi_generic && (act_none || (act_all && !p.resid)) && !(p.gate && !p.resid)) {
        const int rows_valid = p.M - mw0;
        if (rows_valid > 0) {
          const int bA = mw0 / p.cm.rpb;                              // the batch of the wave's first row
          const int split = (bA + 1) * p.cm.rpb - mw0;                 // local row where the next batch starts (>= 128: not in this wave)
          const long long off0 = p.cm.off(mw0);
          const long long off1 = split < 128 && mw0 + split < p.M ? p.cm.off(mw0 + split) - (long long)split * p.cm.ld : off0;
          if (!p.resid) {
            if (act_none) staged_rows_fast8<MI, 0, false, true>(p, off0, bA, nw0, l, acc, scr, rows_valid, split, off1);
            else staged_rows_fast8<MI, 0, true, true>(p, off0, bA, nw0, l, acc, scr, rows_valid, split, off1);
          } else if (p.gate) staged_rows_fast8<MI, 3, false, true>(p, off0, bA, nw0, l, acc, scr, rows_valid, split, off1);
          else staged_rows_fast8<MI, 1, false, true>(p, off0, bA, nw0, l, acc, scr, rows_valid, split, off1);
        }
        return m0 + TM <= p.M;          // every row stored: 32 stores per wave, as in the interior form
      }
    }
  }
  if (m0 + TM <= p.M && n0 + TN <= p.N) {
    staged_rows<MI, TM, false, NI, 0, G0>(p, m0, mw0, n0, nw0, l, acc, scr);
    if constexpr (NI > 4) staged_rows<MI, TM, false, NI, 4, NI - 4>(p, m0, mw0, n0, nw0 + 64, l, acc, scr);
  } else {
    staged_rows<MI, TM, true, NI, 0, G0>(p, m0, mw0, n0, nw0, l, acc, scr);
    if constexpr (NI > 4) staged_rows<MI, TM, true, NI, 4, NI - 4>(p, m0, mw0, n0, nw0 + 64, l, acc, scr);
  }
  return false;
}

// tile selection shared by both kernels: XCD-contiguous, grouped along M for L2 reuse of the W panel
__device__ __forceinline__ void pick_tile(const GemmKArgs& p, int bid, int& tm, int& tn) {
  const int nwg = p.tiles_m * p.tiles_n;
  const int wg = xcd_remap(bid, nwg);
  const int GROUP_M = p.group_m;
  const int in_group = GROUP_M * p.tiles_n;
  const int gid = wg / in_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int rem = wg - gid * in_group;
  tm = first_m + rem % gsz;
  tn = rem / gsz;
}

template <int MODE>  // 0: batched rows, 1: conv3x3 implicit GEMM
__global__ __launch_bounds__(256, 2) void gemm_bf16_t128(GemmKArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES + 4 * 2048];  // A0 A1 B0 B1 + one epilogue slab per wave
  const int w = wave_id();
  const int l = lane_id();
  const int wr = w >> 1, wc = w & 1;

  // ---- tile selection: XCD-contiguous, grouped along M for L2 reuse of the W panel ----
  const int nwg = p.tiles_m * p.tiles_n;
  int wg = xcd_remap((int)blockIdx.x, nwg);
  constexpr int GROUP_M = 8;
  const int in_group = GROUP_M * p.tiles_n;
  const int gid = wg / in_group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int rem = wg - gid * in_group;
  int tm = first_m + rem % gsz;
  const int tn = rem / gsz;
  if (MODE == 0) pick_segment(p, tm);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging addresses: wave w stages 8-row chunks {4w..4w+3} of both tiles ----
  // descriptors are based at the tile's first row (addresses grow with the row index), so operands
  // of any size work with 32-bit in-tile offsets; rows are clamped, so no access leaves the tensor
  const long long a0 = MODE == 0 ? p.am.off(m0) : p.cv.off(m0);
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + a0), 0, 0x7ffffff0u, 0x00020000);
  // W descriptor is based at this tile's first row, so stacked weights of any size work
  const int wrows = min(BN, p.N - n0);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)n0 * p.K), 0,
                                                                 (unsigned)((long long)wrows * p.K * 2), 0x00020000);
  unsigned voffA[4], voffW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (w * 4 + i) * 8 + (l >> 3);           // row within tile
    const int slot = (l & 7) ^ ((row >> 1) & 7);           // logical 16-B slot this lane fetches
    int ra = min(m0 + row, p.M - 1);                       // clamp: rows past the edge are never stored
    int rw = min(row, wrows - 1);
    voffA[i] = (unsigned)(((MODE == 0 ? p.am.off(ra) : p.cv.off(ra)) - a0 + slot * 8) * 2);
    voffW[i] = (unsigned)(((long long)rw * p.K + slot * 8) * 2);
  }

  const int cchunks = MODE == 1 ? p.cv.Cin / BK : 1;
  auto stage = [&](int buf, int kt) {
    const int soff = kt * (BK * 2);
    int soffA = soff;
    if (MODE == 1) {
      const int tap = kt / cchunks, cc = kt - tap * cchunks;
      const int r = tap / 3, sx = tap - r * 3;
      soffA = ((r * p.cv.Wp + sx) * p.cv.Cin + cc * BK) * 2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      DRAG_LDS char* dA = (DRAG_LDS char*)smem + buf * TILE_BYTES + (w * 4 + i) * 1024;
      DRAG_LDS char* dB = (DRAG_LDS char*)smem + (2 + buf) * TILE_BYTES + (w * 4 + i) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (DRAG_LDS void*)dA, 16, voffA[i], soffA, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (DRAG_LDS void*)dB, 16, voffW[i], soff, 0, 0);
    }
  };

  // ---- fragment read addresses (bytes within a tile) ----
  const int p0 = (l >> 4) ^ ((l & 15) >> 1);
  const int fa = (wr * 64 + (l & 15)) * 128;   // + mi*2048, slot (p0 ^ 4ks)*16
  const int fb = (wc * 64 + (l & 15)) * 128;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
    const char* sA = smem + buf * TILE_BYTES;
    const char* sB = smem + (2 + buf) * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ((p0 ^ (ks * 4)) << 4);
      bf16x8_t xa[4], wb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xa[i] = *(const bf16x8_t*)(sA + fa + i * 2048 + so);
        wb[i] = *(const bf16x8_t*)(sB + fb + i * 2048 + so);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ni], xa[mi], acc[mi][ni], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds C[m = .. + (l&15)][n = .. + (l>>4)*4 + 0..3] ----
  const GemmKArgs pd = dest_of(p, n0);
  if (p.wide) staged_epilogue<4, BM>(pd, m0, m0 + wr * 64, n0, n0 + wc * 64, l, acc, smem + 4 * TILE_BYTES + w * 2048);
  else wave_epilogue<4, BM>(pd, m0, m0 + wr * 64 + (l & 15), n0, n0 + wc * 64 + (l >> 4) * 4, acc);
}


// --------------------------------------------------------------------------------------------
// gemm_bf16_deep — (32*MI) x 128 x 64 tile, 4 waves (2x2), wave tile (16*MI) x 64, ST-stage LDS-DMA ring with COUNTED waits.
// For the launches that cannot fill the chip with 256x256 tiles (BASELINE configs[1]: 512 / 1024 / 1536 rows): there a
// workgroup's K loop is a latency chain — the double-buffered t128 loop above exposes one L2/HBM round trip per K-step
// (0.9 us per step measured at K = 12288 / 15360, 28 % of a CU's MFMA rate) — so the ring keeps ST-1 K-steps in flight per
// workgroup and smaller M tiles put more workgroups on the chip.  Same MFMA, same k order per output element as the other
// two kernels: bit-identical results, so the choice may depend on the launch's shape (batch invariance is kept).
// Stage = A tile (32*MI rows) then W tile (128 rows), 128 B per row, same XOR swizzle as t128.  Per stage a wave issues
// MI A chunks... (32*MI / 8 / 4) + 4 W chunks of 8 rows.  The epilogue slabs alias stage memory after the loop's last barrier.
// --------------------------------------------------------------------------------------------
// Round 3: the N extent of the tile is a template parameter too (NI column blocks of 16 per wave: 128- or 192-column tiles).
// A launch of this family is bound by what ONE CU can ingest from L2 (measured ~70 GB/s per CU through LDS-DMA, whatever the
// ring depth): its time is (K-steps) x (tile rows + tile columns) x 128 B x (tiles on the busiest CU) / that rate.  BASELINE
// configs[1]'s two heaviest shapes sit badly on 128-column tiles: (1536, 3072, 15360) is 288 128x128 tiles on 256 CUs (32 CUs
// carry two: 660 TFLOP/s) and (1536, 12288, 3072) is 1152 of them; 96x192 tiles make the first exactly 256 workgroups (one per
// CU, 44 % fewer bytes on the busiest CU) and 128x192 tiles make the second exactly 3 rounds of 256.  Same MFMA, same k order per
// output element: the bits cannot tell (test_gemm_kernels_are_bit_identical), so the choice stays a function of the launch shape.
// one ds_read_b128 the compiler does not see (no automatic s_waitcnt: the caller counts), N of them 2 KiB apart from BASE
template <int OFF>
__device__ __forceinline__ void lds_read_b128(bf16x8_t& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N, int BASE, int I = 0>
__device__ __forceinline__ void lds_read_frags(bf16x8_t* f, unsigned addr) {
  if constexpr (I < N) {
    lds_read_b128<BASE + I * 2048>(f[I], addr);
    lds_read_frags<N, BASE, I + 1>(f, addr);
  }
}

template <int MI, int ST, int NI = 4>
// The ring is DYNAMIC shared memory and the kernel asks for two waves per SIMD: told the static LDS size of a one-workgroup-per-CU ring,
// hipcc sees a lone wave per SIMD, takes its 512-register budget, parks the accumulators in AGPRs and shuttles the loop-carried fragment
// set of the pipelined loop through ~300 v_accvgpr moves per K-step; inside 256 unified registers everything stays in arch VGPRs.
__global__ __launch_bounds__(256, 2) void gemm_bf16_deep(GemmKArgs p) {
  constexpr int TBM = 32 * MI, TBN = 32 * NI;
  constexpr int A_BYTES = TBM * 128, W_BYTES = TBN * 128, STAGE = A_BYTES + W_BYTES;
  constexpr int CA = TBM / 32;                 // A chunks (8 rows, 1 KiB) per wave per stage
  constexpr int CW = TBN / 32;                 // W chunks per wave per stage
  constexpr int CH = CA + CW;                  // DMA instructions per wave per stage
  static_assert(ST >= 2 && ST <= 4 && ST * STAGE >= 4 * 2048 && ST * STAGE <= 160 * 1024 && (NI == 4 || NI == 6), "ring depth / tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];       // ST * STAGE bytes (deep_lds_bytes)
  const int w = wave_id();
  const int l = lane_id();
  const int wr = w >> 1, wc = w & 1;
  int tm, tn;
  pick_tile(p, (int)blockIdx.x, tm, tn);
  pick_segment(p, tm);
  const int m0 = tm * TBM, n0 = tn * TBN;

  const long long a0 = p.am.off(m0);
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + a0), 0, 0x7ffffff0u, 0x00020000);
  const int wrows = min(TBN, p.N - n0);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)n0 * p.K), 0,
                                                                 (unsigned)((long long)wrows * p.K * 2), 0x00020000);
  unsigned voffA[4], voffW[6];      // (an array of template-dependent bound captured by the lambda below loses the host stub in hipcc 7.2)
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int row = (w * CA + i) * 8 + (l >> 3);
    const int slot = (l & 7) ^ ((row >> 1) & 7);
    const int ra = min(m0 + row, p.M - 1);                 // clamp: rows past the edge are never stored
    voffA[i] = (unsigned)((p.am.off(ra) - a0 + slot * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < CW; ++i) {
    const int row = (w * CW + i) * 8 + (l >> 3);
    const int slot = (l & 7) ^ ((row >> 1) & 7);
    const int rw = min(row, wrows - 1);
    voffW[i] = (unsigned)(((long long)rw * p.K + slot * 8) * 2);
  }
  auto stage = [&](int buf, int kt) {
    const int soff = kt * (BK * 2);
    DRAG_LDS char* d = (DRAG_LDS char*)smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < CA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (DRAG_LDS void*)(d + (w * CA + i) * 1024), 16, voffA[i], soff, 0, 0);
#pragma unroll
    for (int i = 0; i < CW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (DRAG_LDS void*)(d + A_BYTES + (w * CW + i) * 1024), 16, voffW[i], soff, 0, 0);
  };

  const int p0 = (l >> 4) ^ ((l & 15) >> 1);
  const int fa = (wr * (TBM / 2) + (l & 15)) * 128;            // + mi*2048
  const int fb = A_BYTES + (wc * (TBN / 2) + (l & 15)) * 128;  // + ni*2048

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- main loop, software-pipelined ACROSS the barrier.  One workgroup per CU means one wave per SIMD, all four in the same phase:
  // with "barrier | ds_read | MFMA" per K-step the LDS phase (72 KiB of fragment reads per K-step for a 96x192 tile = the MFMA time)
  // and the MFMA phase never overlap — (1536, 3072, 15360) ran 167 us where its operand stream alone takes 96 us and its MFMAs 58
  // (scripts/probe/probe_ingest.hip).  So the fragments live in two register sets (a lone wave per SIMD has 512 VGPRs): the k-half-1
  // reads of K-step kt issue before its k-half-0 MFMAs, the barrier of K-step kt+1 sits BETWEEN the two MFMA halves, and the k-half-0
  // reads of K-step kt+1 issue before the k-half-1 MFMAs of kt.  At that barrier every wave has finished reading buffer kt, so K-step
  // kt+ST is staged into it: ST K-steps in flight instead of ST-1 from the same LDS.  Same MFMA order per accumulator: same bits.
  const int nk = p.K / BK;
#pragma unroll
  for (int s2 = 0; s2 < ST; ++s2)
    if (s2 < nk) stage(s2, s2);
  auto wait_landed = [&](int younger) {      // the K-step awaited has `younger` stages behind it in this wave's (in-order) VMEM queue
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CH) : "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CH) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // The fragment reads are inline asm and the waits that retire them are counted by hand: left to the compiler, the older register set
  // is awaited with lgkmcnt(0) — which also drains the reads just issued for the other set, i.e. no overlap at all.  LDS returns in
  // order, so lgkmcnt(MI + NI) after issuing one set's reads means the previous set has landed.
  bf16x8_t xa0[MI], wb0[NI], xa1[MI], wb1[NI];
  const unsigned lds0 = (unsigned)(size_t)(DRAG_LDS char*)smem;
  const unsigned adA = lds0 + (unsigned)fa, adB = lds0 + (unsigned)(fb - A_BYTES);     // + buffer * STAGE + k-half slot
  auto read_half = [&](int b, int ks, bf16x8_t* xa, bf16x8_t* wb) {
    const unsigned so = (unsigned)(b * STAGE + ((p0 ^ (ks * 4)) << 4));
    lds_read_frags<MI, 0>(xa, adA + so);
    lds_read_frags<NI, A_BYTES>(wb, adB + so);
  };
  auto landed = [&](bf16x8_t* xa, bf16x8_t* wb) {      // after a wait: what was read into these registers may be used from here on
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(xa[i]));
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(wb[i]));
  };
  wait_landed(min(ST - 1, nk - 1));
  // A bare s_barrier: __syncthreads() carries a release fence, i.e. s_waitcnt vmcnt(0), which would drain the ring.
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_half(0, 0, xa0, wb0);
  int buf = 0;
#define DRAG_DEEP_MMA(XA, WB) _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) \
    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WB[ni], XA[mi], acc[mi][ni], 0, 0, 0)
  int kt = 0;
  // steady state: K-steps kt+1 .. kt+ST-1 are issued and K-step kt+ST exists — one basic block per K-step, nothing conditional
  for (; kt + ST < nk; ++kt) {
    const int nb = buf + 1 == ST ? 0 : buf + 1;
    read_half(buf, 1, xa1, wb1);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MI + NI) : "memory");      // k-half 0 of this K-step (read one MFMA half ago) landed
    landed(xa0, wb0);
    DRAG_DEEP_MMA(xa0, wb0);
    __builtin_amdgcn_sched_barrier(0);                           // (the waits below must not rise above the MFMAs)
    // K-step kt+1 landed for this wave (ST-2 younger stages stay in flight) and its own reads of buffer kt are complete ...
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((ST - 2) * CH) : "memory");
    // ... for every wave
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stage(buf, kt + ST);
    read_half(nb, 0, xa0, wb0);
    landed(xa1, wb1);                      // (volatile asm keeps its order: this half's MFMAs cannot rise above the reads just issued)
    DRAG_DEEP_MMA(xa1, wb1);
    buf = nb;
  }
  // the last ST K-steps: nothing left to stage, the ring drains
  for (; kt < nk; ++kt) {
    const int nb = buf + 1 == ST ? 0 : buf + 1;
    read_half(buf, 1, xa1, wb1);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MI + NI) : "memory");
    landed(xa0, wb0);
    DRAG_DEEP_MMA(xa0, wb0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) {
      wait_landed(min(ST - 2, nk - 2 - kt));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_half(nb, 0, xa0, wb0);
      landed(xa1, wb1);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      landed(xa1, wb1);
    }
    DRAG_DEEP_MMA(xa1, wb1);
    buf = nb;
  }
#undef DRAG_DEEP_MMA
  const GemmKArgs pd = dest_of(p, n0);
  if (p.wide) {
    __syncthreads();                              // the slabs alias the ring
    staged_epilogue<MI, TBM, TBN, NI>(pd, m0, m0 + wr * (TBM / 2), n0, n0 + wc * (TBN / 2), l, acc, smem + w * 2048);
  } else {
    wave_epilogue<MI, TBM, TBN, NI>(pd, m0, m0 + wr * (TBM / 2) + (l & 15), n0, n0 + wc * (TBN / 2) + (l >> 4) * 4, acc);
  }
}

// --------------------------------------------------------------------------------------------
// gemm_bf16_w4 (round 5; EXPERIMENT, DRAG_EXPERIMENTS builds only: "gemm_kernel" = 400 + V) — the 256x256x64 tile as FOUR waves x (128 x 128),
// one wave per SIMD with the whole register file (256 accumulators in AGPRs): a third less LDS -> register traffic per flop than the
// 8-wave kernel below, the shape of the vendor library's kernel on this chip.  hipcc cannot schedule a 512-register wave
// (gemm_bf16_deep<8, 2, 8>: waterfall loops around every LDS-DMA, 168 v_accvgpr moves per K-step), so the K loop is ONE asm statement whose
// text scripts/gen/gemm4w_kloop.py generates (register map and schedule there); the kernel binds its operands to the physical registers
// the text names.  Plain tiles (not persistent), K a multiple of 128.  Bit-identical to every other GEMM kernel here.
// MEASURED (profiles/r05_gemm_w4_*.log; us per K-step and tile round, the 8-wave kernel 1.45-1.49 on the same boxes): V1 = refill by
// LDS-DMA 1.50-1.53, V0 = refill through registers (every chunk a full K-step in flight) 1.58; ablations: no refill 1.12-1.15 (= 2048
// MFMA cycles at 96 %: the MFMA + fragment-read skeleton is fine), no barrier -0.07...-0.16, every chunk from an L2-resident K-step 1.26.
// So 0.3 us of a K-step is the operand stream pushing back on the ISSUE of a lone wave's loads (not latency: a K-step of flight per chunk
// does not help) — exactly what the 8-wave kernel's second wave group hides.  Not the product kernel.
// --------------------------------------------------------------------------------------------
#include "gemm4w_kloop.h"
typedef __attribute__((ext_vector_type(32))) float f32x32_t;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8_t;

#if DRAG_EXP
template <int V>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4(GemmKArgs p) {
  constexpr int A_BYTES = 256 * 128, STAGE = 2 * A_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];       // 2 * STAGE = 128 KiB
  const int w = wave_id();
  const int l = lane_id();
  const int wr = w >> 1, wc = w & 1;
  int tm, tn;
  pick_tile(p, (int)blockIdx.x, tm, tn);
  pick_segment(p, tm);
  const int m0 = tm * 256, n0 = tn * 256;
  const long long a0 = p.am.off(m0);
  const int wrows = min(256, p.N - n0);
  // descriptors as four dwords each (base, base_hi, num_records, flags): operands of the asm statement
  const unsigned long long pa = (unsigned long long)(uintptr_t)(p.A + a0), pw = (unsigned long long)(uintptr_t)(p.W + (long long)n0 * p.K);
  auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };      // wave-uniform by construction: say so
  // bounded: the register form requests up to one K-step past K in its last iterations (zeros / the next row's start, never used)
  const long long a_span = (p.am.off(min(m0 + 255, p.M - 1)) - a0 + p.K) * 2;
  const u32x4_t rsA = {uni((uint32_t)pa), uni((uint32_t)(pa >> 32) & 0xffffu), uni((uint32_t)a_span), 0x00020000u};
  const u32x4_t rsW = {uni((uint32_t)pw), uni((uint32_t)(pw >> 32) & 0xffffu), uni((uint32_t)((long long)wrows * p.K * 2)), 0x00020000u};
  u32x8_t voA, voW;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (w * 8 + i) * 8 + (l >> 3);
    const int slot = (l & 7) ^ ((row >> 1) & 7);
    const int ra = min(m0 + row, p.M - 1);                 // clamp: rows past the edge are never stored
    voA[i] = (unsigned)((p.am.off(ra) - a0 + slot * 8) * 2);
    const int rw = min(row, wrows - 1);
    voW[i] = (unsigned)(((long long)rw * p.K + slot * 8) * 2);
  }
  const unsigned lds0 = (unsigned)(size_t)(DRAG_LDS char*)smem;
  const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)w * 8192u));
  const int p0 = (l >> 4) ^ ((l & 15) >> 1);
  const int fa = (wr * 128 + (l & 15)) * 128;
  const int fb = A_BYTES + (wc * 128 + (l & 15)) * 128;
  u32x8_t rd;      // [buffer][X k-half 0, X k-half 1, W k-half 0, W k-half 1]
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      rd[4 * b + ks] = lds0 + (unsigned)(b * STAGE + fa + ((p0 ^ (ks * 4)) << 4));
      rd[4 * b + 2 + ks] = lds0 + (unsigned)(b * STAGE + fb + ((p0 ^ (ks * 4)) << 4));
    }
  f32x32_t accrow[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 32; ++r) accrow[i][r] = 0.f;
  unsigned n2 = (unsigned)(p.K / 128 - 1);       // pairs of K-steps in the steady loop; the last pair is the tail
  unsigned soff = 0;
  // prologue: K-steps 0 and 1 into the two stage buffers, then K-step 0 visible to every wave
  const u32x2_t wrv = {lds0 + (unsigned)(w * 8192 + l * 16), lds0 + (unsigned)(STAGE + w * 8192 + l * 16)};
#define G4W_OUTS                                                                                                                              \
  "+{a[0:31]}"(accrow[0]), "+{a[32:63]}"(accrow[1]), "+{a[64:95]}"(accrow[2]), "+{a[96:127]}"(accrow[3]), "+{a[128:159]}"(accrow[4]),        \
      "+{a[160:191]}"(accrow[5]), "+{a[192:223]}"(accrow[6]), "+{a[224:255]}"(accrow[7]), [n2] "+s"(n2), [soff] "+s"(soff)
#define G4W_INS "{v[128:135]}"(voA), "{v[136:143]}"(voW), "{v[144:151]}"(rd), [rsa] "s"(rsA), [rsw] "s"(rsW), [ldsw] "s"(ldsw)
  if constexpr (V == 1) {          // form D: refill by LDS-DMA
    asm volatile(G4W_D_STAGE0 G4W_FIRST_READS G4W_D_LOOP G4W_D_TAIL : G4W_OUTS : G4W_INS : G4W_CLOBBERS, "scc", "memory");
  } else if constexpr (V == 2) {   // form D2: LDS-DMA, two barriers per K-step, the refill spread from the first barrier on
    asm volatile(G4W_D_STAGE0 G4W_FIRST_READS G4W_D2_LOOP G4W_D_TAIL : G4W_OUTS : G4W_INS : G4W_CLOBBERS, "scc", "memory");
  } else if constexpr (V == 3) {
    asm volatile(G4W_D_STAGE0 G4W_FIRST_READS G4W_D2_LOOP_A G4W_D_TAIL : G4W_OUTS : G4W_INS : G4W_CLOBBERS, "scc", "memory");
  } else if constexpr (V == 4) {
    asm volatile(G4W_D_STAGE0 G4W_FIRST_READS G4W_D2_LOOP_B G4W_D_TAIL : G4W_OUTS : G4W_INS : G4W_CLOBBERS, "scc", "memory");
  } else if constexpr (V == 5) {
    asm volatile(G4W_D_STAGE0 G4W_FIRST_READS G4W_D2_LOOP_C G4W_D_TAIL : G4W_OUTS : G4W_INS : G4W_CLOBBERS, "scc", "memory");
  } else {                         // form R: refill through registers
    asm volatile(G4W_R_STAGE0 G4W_FIRST_READS G4W_R_LOOP G4W_R_TAIL : G4W_OUTS : G4W_INS, "{v[152:153]}"(wrv) : G4W_CLOBBERS_R, "scc", "memory");
  }
#undef G4W_OUTS
#undef G4W_INS
  f32x4_t acc[8][8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = accrow[mi][4 * ni + r];
  const GemmKArgs pd = dest_of(p, n0);
  if (p.wide) {
    __syncthreads();                              // the slabs alias the stage buffers
    staged_epilogue<8, 256, 256, 8>(pd, m0, m0 + wr * 128, n0, n0 + wc * 128, l, acc, smem + w * 4096);
  } else {
    wave_epilogue<8, 256, 256, 8>(pd, m0, m0 + wr * 128 + (l & 15), n0, n0 + wc * 128 + (l >> 4) * 4, acc);
  }
}

#endif

// gemm_bf16_w4p — PRODUCT kernel of the large Linears since round 5: the persistent form of gemm_bf16_w4 (form D2 of the K loop): one workgroup per CU walks tiles b, b + P, ... (all on its
// XCD); a tile is ONE asm statement (K-steps 0 and 1 already staged, steady loop, a tail whose two K-steps stage K-steps 0 and 1 of the
// workgroup's next tile), then the C++ epilogue on slabs that do not alias the stage buffers — so the epilogue overlaps the next tile's
// loads.  K a multiple of 128; batched rows / two destinations / every epilogue form like gemm_bf16_t256<0>; no conv mode, no pair.
typedef __attribute__((ext_vector_type(16))) uint32_t u32x16_t;
struct W4Tile {
  u32x4_t rsA, rsW;
  u32x16_t vo;      // [0:7] X chunks, [8:15] W chunks
  int m0, n0;       // the tile's first row / column (wave-uniform, in SGPRs: the epilogue of the tile reuses them instead of walking again)
};
// An INTERIOR tile inside one batch of the row map has offsets row * ld (no clamp, no division); edge tiles and tiles that cross a batch
// take the general form
__device__ __forceinline__ void w4_tile_state(const GemmKArgs& p, int tile, int w, int l, bool valid, W4Tile& t) {
  auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  int tm, tn;
  pick_tile(p, tile, tm, tn);
  const int m0 = __builtin_amdgcn_readfirstlane(tm * 256), n0 = __builtin_amdgcn_readfirstlane(tn * 256);
  t.m0 = m0;
  t.n0 = n0;
  long long a0 = p.am.off(m0);
  const int wrows = min(256, p.N - n0);
  const bf16_t *Ab = p.A, *Wb = p.W;
  if (p.w_boff) {                            // split-K: row batch = K slice; in a pair's launch the rows behind split_m1 are the second problem's
    const int sl = m0 / p.am.rpb, r = m0 - sl * p.am.rpb;
    Wb += (long long)sl * p.w_boff;
    if (p.split_m1 > 0 && r >= p.split_m1) {
      Ab = p.A2; Wb = p.W2 + (long long)sl * p.w_boff;
      a0 = (long long)sl * p.am.bs + (long long)(r - p.split_m1) * p.am.ld;
    }
  }
  const unsigned long long pa = (unsigned long long)(uintptr_t)(Ab + a0), pw = (unsigned long long)(uintptr_t)(Wb + (long long)n0 * p.ldw);
  const bool interior = m0 + 256 <= p.M && wrows == 256 && m0 / p.am.rpb == (m0 + 255) / p.am.rpb;
  const long long a_span = interior ? ((long long)255 * p.am.ld + p.K) * 2 : (p.am.off(min(m0 + 255, p.M - 1)) - a0 + p.K) * 2;
  // no next tile: descriptors with zero records — the tail's loads return zeros without touching memory
  t.rsA = (u32x4_t){uni((uint32_t)pa), uni((uint32_t)(pa >> 32) & 0xffffu), valid ? uni((uint32_t)a_span) : 0u, 0x00020000u};
  t.rsW = (u32x4_t){uni((uint32_t)pw), uni((uint32_t)(pw >> 32) & 0xffffu), valid ? uni((uint32_t)(((long long)(wrows - 1) * p.ldw + p.K) * 2)) : 0u, 0x00020000u};
  if (interior) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (w * 8 + i) * 8 + (l >> 3);
      const int slot = (l & 7) ^ ((row >> 1) & 7);
      t.vo[i] = (unsigned)((row * p.am.ld + slot * 8) * 2);
      t.vo[8 + i] = (unsigned)((row * p.ldw + slot * 8) * 2);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (w * 8 + i) * 8 + (l >> 3);
    const int slot = (l & 7) ^ ((row >> 1) & 7);
    const int ra = min(m0 + row, p.M - 1);                 // clamp: rows past the edge are never stored
    t.vo[i] = (unsigned)((p.am.off(ra) - a0 + slot * 8) * 2);
    const int rw = min(row, wrows - 1);
    t.vo[8 + i] = (unsigned)(((long long)rw * p.ldw + slot * 8) * 2);
  }
}

#if DRAG_EXP
// experiment builds: shader-clock stamps of workgroup 0 / wave 0 around the pieces of a tile (drag_debug_w4_stamps copies them out)
__device__ unsigned long long g_w4_stamps[8 * 64];
#define W4_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if (blockIdx.x == 0 && w == 0 && l == 0 && tile_no < 64) g_w4_stamps[tile_no * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define W4_STAMP(slot) do { } while (0)
#endif

__global__ __launch_bounds__(256, 1) void gemm_bf16_w4p(GemmKArgs p) {
  constexpr int A_BYTES = 256 * 128, STAGE = 2 * A_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];       // 2 * STAGE + 4 x 2 epilogue slabs of 2 KiB
  const int w = wave_id();
  const int l = lane_id();
  const int wr = w >> 1, wc = w & 1;
  const int P = (int)gridDim.x;
  const int nwg = p.tiles_m * p.tiles_n;
  int vb = (int)blockIdx.x;
  const unsigned lds0 = (unsigned)(size_t)(DRAG_LDS char*)smem;
  const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)w * 8192u));
  // Per-lane constants are RECOMPUTED from a laundered lane id in every tile (a handful of VALU instructions): hoisted out of the tile loop
  // they are live across the K loop's statement, which leaves the compiler 88 free VGPRs (v0-v127 clobbered, v128-v151 / v224-v239 bound) —
  // it spilled them to scratch, and every reload is an s_waitcnt vmcnt(0) that drains the previous epilogue's 32 stores before the next K
  // loop may start (stamps: 2800 cycles of "tile state" per tile, all of it that wait)
  auto lane_now = [&]() { int v = l; asm volatile("" : "+v"(v)); return v; };
  // the same for the divisors of the tile walk and the row maps: the reciprocals of wave-uniform divisions are computed by the VALU, and
  // hoisted they sit in VGPRs across the statement
  auto args_now = [&]() {
    GemmKArgs q = p;
    asm volatile("" : "+s"(q.tiles_n), "+s"(q.tiles_m), "+s"(q.group_m), "+s"(q.am.rpb), "+s"(q.cm.rpb), "+s"(q.M), "+s"(q.K), "+s"(q.am.ld), "+s"(q.cm.ld), "+s"(q.ldw));
    return q;
  };
  auto read_addrs = [&](int lv) {      // [buffer][X k-half 0, X k-half 1, W k-half 0, W k-half 1]
    const int p0 = (lv >> 4) ^ ((lv & 15) >> 1);
    const int fa = (wr * 128 + (lv & 15)) * 128;
    const int fb = A_BYTES + (wc * 128 + (lv & 15)) * 128;
    u32x8_t rd;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        rd[4 * b + ks] = lds0 + (unsigned)(b * STAGE + fa + ((p0 ^ (ks * 4)) << 4));
        rd[4 * b + 2 + ks] = lds0 + (unsigned)(b * STAGE + fb + ((p0 ^ (ks * 4)) << 4));
      }
    return rd;
  };
  W4Tile cur, nxt;
  w4_tile_state(p, vb, w, lane_now(), true, cur);
  unsigned soff = 0;
  // the workgroup's first tile: K-steps 0 and 1 into the two stage buffers (every later tile finds them staged by its predecessor's tail)
  asm volatile(G4W_D_STAGE0_NOWAIT : [soff] "+s"(soff)
               : "{v[128:143]}"(cur.vo), [rsa] "s"(cur.rsA), [rsw] "s"(cur.rsW), [ldsw] "s"(ldsw) : "scc", "m0", "memory");
  int stores_behind = 0;
  [[maybe_unused]] int tile_no = 0;
  const bool late = DRAG_EXP && p.w4_late_state != 0;      // (experiment builds only: the product kernel must not carry the variant's 64 B of scratch)
  if (late) w4_tile_state(args_now(), vb + P < nwg ? vb + P : vb, w, lane_now(), vb + P < nwg, nxt);
  for (;;) {
    W4_STAMP(0);
    const bool have_next = vb + P < nwg;
    const int lt = lane_now();
    if (!late) w4_tile_state(args_now(), have_next ? vb + P : vb, w, lt, have_next, nxt);
    const u32x8_t rd = read_addrs(lt);
    W4_STAMP(1);
    // K-steps 0 and 1 of this tile landed (this wave's pieces; the statement below opens with the barrier).  Behind an interior tile's fast
    // epilogue exactly 32 stores are younger than those pieces (VMEM operations of a wave retire in issue order): they may stay in flight
    if (stores_behind == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4_STAMP(2);
    f32x32_t accrow[8];                            // written by the statement (the first K-step's MFMAs start from the constant 0)
    unsigned n2 = (unsigned)(p.K / 128 - 2);       // pairs of K-steps in the steady loop: all but the first pair and the tail
    asm volatile(G4W_P_FIRST G4W_P_PAIR0 G4W_P_LOOP G4W_P_TAIL
                 : "={a[0:31]}"(accrow[0]), "={a[32:63]}"(accrow[1]), "={a[64:95]}"(accrow[2]), "={a[96:127]}"(accrow[3]),
                   "={a[128:159]}"(accrow[4]), "={a[160:191]}"(accrow[5]), "={a[192:223]}"(accrow[6]), "={a[224:255]}"(accrow[7]),
                   [n2] "+s"(n2), [soff] "+s"(soff)
                 : "{v[128:143]}"(cur.vo), "{v[224:239]}"(nxt.vo), "{v[144:151]}"(rd), [rsa] "s"(cur.rsA), [rsw] "s"(cur.rsW),
                   [rsa2] "s"(nxt.rsA), [rsw2] "s"(nxt.rsW), [ldsw] "s"(ldsw)
                 : G4W_CLOBBERS, "scc", "memory");
    W4_STAMP(3);
    W4Tile nn;
    if (late) {            // (measurement) the state of the tile after next, in front of this tile's epilogue
      const bool have2 = vb + 2 * P < nwg;
      w4_tile_state(args_now(), have2 ? vb + 2 * P : vb, w, lane_now(), have2, nn);
    }
    const int le = lane_now();
    f32x4_t acc[8][8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = accrow[mi][4 * ni + r];
    const GemmKArgs pe = args_now();
    const int m0 = cur.m0, n0 = cur.n0;
    const GemmKArgs pd = dest_of(pe, n0);
    bool fast = false;
    if (pd.wide) fast = staged_epilogue<8, 256, 256, 8>(pd, m0, m0 + wr * 128, n0, n0 + wc * 128, le, acc, smem + 2 * STAGE + w * 4096);
    else wave_epilogue<8, 256, 256, 8>(pd, m0, m0 + wr * 128 + (le & 15), n0, n0 + wc * 128 + (le >> 4) * 4, acc);
    stores_behind = fast ? 32 : 0;
    W4_STAMP(4);
    ++tile_no;
    if (!have_next) break;
    cur = nxt;
    if (late) nxt = nn;
    vb += P;
  }
}

// --------------------------------------------------------------------------------------------
// gemm_bf16_t256 — 256x256x64 tile, 8 waves (2 along M x 4 along N), wave tile 128x64 as 8x4
// v_mfma_f32_16x16x32_bf16 (128 accumulator registers).  LDS: 2 K-tile buffers x {A0,A1,B0,B1}
// half-tiles of 128 rows x 64 k (16 KiB each) = 128 KiB, one workgroup per CU, 2 waves per SIMD.
//
// Per K-tile t (buffer t&1) TWO phases, each = load segment | barrier | 32 MFMAs | barrier (4 barriers per K-tile):
//   phase A  reads W cols 0-63 + X rows 0-63 (16 x ds_read_b128)   quadrants (0,0) (0,1)
//   phase B  reads X rows 64-127 (8)                               quadrants (1,1) (1,0)
// A wave never needs a whole K-tile at once, so the LDS-DMA stream is cut into four 16-KiB PIECES ordered by
// need-time instead of by operand:
//   alpha = A rows 0-63 of both halves          beta  = W rows {0-31, 64-95} of both halves       (read in A)
//   gamma = W rows {32-63, 96-127}   (read in A) delta = A rows 64-127 of both halves             (read in B)
// and issued into the slot whose last reader finished >= 1 phase earlier:
//   phase A(t): delta(t+1)   (2 DMA per wave)        phase B(t): alpha, beta, gamma (t+2)   (6 DMA per wave)
// -> load segments of 16 reads + 2 DMA and 8 reads + 6 DMA, both shorter than the partner group's 32-MFMA segment;
// every piece is in flight for 2 phases (one K-tile) before the wait that retires it, ~80 KiB are in flight per CU
// and the queue is never drained: both waits are the COUNTED s_waitcnt vmcnt(8) (four younger pieces stay in flight).
// (Measured alternatives, same data: four phases of 16 MFMAs with 8 barriers per K-tile -3 %; DMA issue inside the
//  MFMA segment -10 %; k-step-split fragment reads -3 %; 32x32x16 MFMA -7 %; waiting for reads after the barrier +-0.)
// The two wave groups (wr = 0 / 1: one wave of each per SIMD) run staggered by one barrier, so one group's MFMA
// segment overlaps the other's ds_read / DMA-issue segment (s_setprio favours the MFMA side).
// Hazard rules this schedule satisfies: (RAW) data read in the load segment of phase p is waited for (vmcnt) by
// EVERY wave in the load segment of phase p-1, i.e. before a barrier that the staggered group has passed before
// it reads; (WAR) every ds_read is retired (lgkmcnt(0)) before its phase's first barrier and a slot is restaged
// >= 1 phase after its last read; the compiler may not move anything across a barrier (sched_barrier).
// --------------------------------------------------------------------------------------------
constexpr int T2_HALF = 128 * BK * 2;          // 16 KiB half-tile
constexpr int T2_BUF = 4 * T2_HALF;            // A0 A1 B0 B1

#define T2_BARRIER()                      \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

template <int MODE, bool SEG>   // SEG: the launch may carry a second row segment (drag_gemm_bf16_pair)
__device__ __forceinline__ void t256_body(const GemmKArgs& p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * T2_BUF + 8 * 2048];   // + one 2 KiB epilogue slab per wave
  const int w = wave_id();
  const int l = lane_id();
  const int wr = w >> 2, wc = w & 3;
  // ---- persistent: this workgroup computes tiles vb, vb + P, vb + 2P ... (P = gridDim.x, a multiple of 8 whenever a
  // workgroup has more than one tile, so every tile of a workgroup maps to the XCD the workgroup runs on).  The LDS-DMA
  // stream runs CONTINUOUSLY across tile boundaries: the last two K-steps of a tile already fetch K-steps 0 and 1 of
  // the next one, so the epilogue's stores overlap the next tile's loads and only the first tile pays a prologue.
  const int P = (int)gridDim.x;
  const int nwg = p.tiles_m * p.tiles_n;
  int vb = (int)blockIdx.x;

  // ---- load state of ONE tile (switched in place two K-steps before the tile's first MFMA)
  __amdgpu_buffer_rsrc_t rsA, rsW;
  unsigned vo[4][2];          // [piece: 0 alpha, 1 beta, 2 gamma, 3 delta][chunk] global byte offset (per lane)
  int lo[4][2];               // LDS byte offset of the chunk inside a K-tile buffer (wave-uniform, tile-independent)
  // staging role: every piece has 16 chunks of 8 rows; this wave moves chunks c = 2w, 2w+1 of each piece.
  // chunk c -> operand half (c>>3) and an 8-row group inside it:
  //   alpha: rows 8*(c&7)              delta: rows 64 + 8*(c&7)
  //   beta : sub=c&7: rows 8*sub (sub<4) | 64 + 8*(sub-4)      gamma: rows 32 + 8*sub | 96 + 8*(sub-4)
  auto chunk_row0 = [&](int pc, int sub) {
    if (pc == 0) return 8 * sub;
    if (pc == 3) return 64 + 8 * sub;
    if (pc == 1) return sub < 4 ? 8 * sub : 64 + 8 * (sub - 4);
    return sub < 4 ? 32 + 8 * sub : 96 + 8 * (sub - 4);
  };
#pragma unroll
  for (int pc = 0; pc < 4; ++pc)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const int c = w * 2 + c2, half = c >> 3;
      lo[pc][c2] = ((pc == 0 || pc == 3 ? 0 : 2) + half) * T2_HALF + chunk_row0(pc, c & 7) * 128;
    }
  auto load_state = [&](int tile) {
    int tm, tn;
    pick_tile(p, tile, tm, tn);
    const bf16_t* A = p.A;
    const bf16_t* W = p.W;
    int M = p.M;
    RowMap am = p.am;
    if (SEG && p.seg_tiles_m > 0 && tm >= p.seg_tiles_m) { tm -= p.seg_tiles_m; A = p.A2; W = p.W2; M = p.M2; am = p.am2; }
    const int m0 = tm * 256, n0 = tn * 256;
    // descriptors are based at the tile's first row, so operands of any size work with 32-bit in-tile offsets
    const long long a0 = MODE == 0 ? am.off(m0) : p.cv.off(m0);
    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + a0), 0, 0x7ffffff0u, 0x00020000);
    const int wrows = min(256, p.N - n0);
    rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long long)n0 * p.K), 0, (unsigned)((long long)wrows * p.K * 2),
                                            0x00020000);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const int c = w * 2 + c2, half = c >> 3;
        const int row = chunk_row0(pc, c & 7) + (l >> 3);      // this lane's row inside the half
        const int slot = (l & 7) ^ ((row >> 1) & 7);
        if (pc == 0 || pc == 3) {
          const int ra = min(m0 + half * 128 + row, M - 1);    // clamp: rows past the edge are never stored
          vo[pc][c2] = (unsigned)(((MODE == 0 ? am.off(ra) : p.cv.off(ra)) - a0 + slot * 8) * 2);
        } else {
          const int rw = min(half * 128 + row, wrows - 1);
          vo[pc][c2] = (unsigned)(((long long)rw * p.K + slot * 8) * 2);
        }
      }
  };
  const int cchunks = MODE == 1 ? p.cv.Cin / BK : 1;
  const int nk = p.K / BK;                          // >= 4 (use_t256)
  auto issue = [&](int pc, int kt, int buf) {       // K-step kt of the tile in the load state -> LDS buffer buf
    int soff = kt * (BK * 2);
    if (MODE == 1 && (pc == 0 || pc == 3)) {
      const int tap = kt / cchunks, cc = kt - tap * cchunks;
      const int r = tap / 3, sx = tap - r * 3;
      soff = ((r * p.cv.Wp + sx) * p.cv.Cin + cc * BK) * 2;
    }
    DRAG_LDS char* d = (DRAG_LDS char*)smem + buf * T2_BUF;
    if (pc == 0 || pc == 3) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (DRAG_LDS void*)(d + lo[pc][0]), 16, vo[pc][0], soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (DRAG_LDS void*)(d + lo[pc][1]), 16, vo[pc][1], soff, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (DRAG_LDS void*)(d + lo[pc][0]), 16, vo[pc][0], soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (DRAG_LDS void*)(d + lo[pc][1]), 16, vo[pc][1], soff, 0, 0);
    }
  };

  // fragment read offsets inside this wave's A half (wr) and B half (wc>>1)
  const int p0 = (l >> 4) ^ ((l & 15) >> 1);
  const int fx = wr * T2_HALF + (l & 15) * 128;                                   // + mi*2048
  const int fw = (2 + (wc >> 1)) * T2_HALF + ((wc & 1) * 64 + (l & 15)) * 128;    // + ni*2048

  // ---- TWO phases per K-step (32 MFMAs each), 4 barriers per K-step:
  //   phase A: reads W cols 0-63 + X rows 0-63 (16 x b128), issues delta(g+1),            quadrants (0,0) (0,1)
  //   phase B: reads X rows 64-127 (8),                     issues alpha,beta,gamma(g+2), quadrants (1,1) (1,0)
  // (16 reads + 2 DMA | 8 reads + 6 DMA: both load segments fit under the partner group's 32-MFMA segment.)
  // stream:  B(g): a,b,g(g+2)   A(g+1): d(g+2)   B(g+1): a,b,g(g+3) ...   every wait leaves 4 younger pieces: vmcnt(8).
  // g counts K-steps over ALL tiles of this workgroup (buffer = g & 1).
  load_state(vb);
  issue(0, 0, 0); issue(1, 0, 0); issue(2, 0, 0); issue(3, 0, 0);
  issue(0, 1, 1); issue(1, 1, 1); issue(2, 1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // alpha, beta, gamma (0) landed
  T2_BARRIER();
  if (wr == 1) T2_BARRIER();                       // stagger the second wave group by one barrier

  bf16x8_t xf[4][2], w0[2][2], w1[2][2];
  f32x4_t acc[8][4];
#define T2_MMA(wsel, mh, nh) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) \
      acc[4 * (mh) + mi][2 * (nh) + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsel[ni][ks], xf[mi][ks], \
                                                                                  acc[4 * (mh) + mi][2 * (nh) + ni], 0, 0, 0)
  // wait until at most `8 + extra` VMEM operations are outstanding; with nothing younger in the stream: drain.
  // Right after an interior tile's epilogue the >= 16 stores it issued sit between the piece waited for and the
  // youngest pieces (VMEM operations of a wave retire in issue order), so 16 more may stay in flight.
#define T2_WAIT(more, relaxed) do { if (!(more)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
                                    else if (relaxed) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); \
                                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); } while (0)

  int g = 0;
  bool after_interior_epilogue = false;
  for (;;) {
    const bool have_next = vb + P < nwg;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nk; ++t, ++g) {
      const char* sb = smem + (g & 1) * T2_BUF;
      const bool more1 = t + 1 < nk || have_next;      // K-step g+1 exists
      const bool more2 = t + 2 < nk || have_next;      // K-step g+2 exists
      const bool relaxed = after_interior_epilogue && t == 0;
      // ================= phase A =================
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          w0[ni][ks] = *(const bf16x8_t*)(sb + fw + ni * 2048 + ((p0 ^ (ks * 4)) << 4));
          w1[ni][ks] = *(const bf16x8_t*)(sb + fw + (2 + ni) * 2048 + ((p0 ^ (ks * 4)) << 4));
        }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xf[mi][ks] = *(const bf16x8_t*)(sb + fx + mi * 2048 + ((p0 ^ (ks * 4)) << 4));
      if (more1) issue(3, t + 1 < nk ? t + 1 : 0, (g + 1) & 1);   // delta(g+1): A rows 64-127 of the other buffer, last read in B(g-1)
      T2_WAIT(more1, relaxed);                          // delta(g) landed (read in phase B)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      T2_BARRIER();
      __builtin_amdgcn_s_setprio(1);
      T2_MMA(w0, 0, 0);
      T2_MMA(w1, 0, 1);
      __builtin_amdgcn_s_setprio(0);
      T2_BARRIER();
      // ================= phase B =================
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xf[mi][ks] = *(const bf16x8_t*)(sb + fx + (4 + mi) * 2048 + ((p0 ^ (ks * 4)) << 4));
      // every later load of this tile has been issued: from here on the stream fetches the next tile
      if (t == nk - 2 && have_next) load_state(vb + P);
      if (more2) {                                     // slots last read in phase A of this K-step
        const int kt = t + 2 < nk ? t + 2 : t + 2 - nk;
        issue(0, kt, g & 1); issue(1, kt, g & 1); issue(2, kt, g & 1);
      }
      T2_WAIT(more2, relaxed);                          // alpha, beta, gamma (g+1) landed (read in A of the next K-step)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      T2_BARRIER();
      __builtin_amdgcn_s_setprio(1);
      T2_MMA(w1, 1, 1);
      T2_MMA(w0, 1, 0);
      __builtin_amdgcn_s_setprio(0);
      T2_BARRIER();
    }
    if (!have_next && wr == 0) T2_BARRIER();         // balance the stagger before the last epilogue
    int tm, tn;
    pick_tile(p, vb, tm, tn);
    GemmKArgs pd = p;
    if (SEG) pick_segment(pd, tm);
    const int m0 = tm * 256, n0 = tn * 256;
    pd = dest_of(pd, n0);
    if (pd.wide) staged_epilogue<8, 256>(pd, m0, m0 + wr * 128, n0, n0 + wc * 64, l, acc, smem + 2 * T2_BUF + w * 2048);
    else wave_epilogue<8, 256>(pd, m0, m0 + wr * 128 + (l & 15), n0, n0 + wc * 64 + (l >> 4) * 4, acc);
    if (!have_next) break;
    after_interior_epilogue = m0 + 256 <= pd.M && n0 + 256 <= p.N;
    vb += P;
  }
#undef T2_MMA
#undef T2_WAIT
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t256(GemmKArgs p) { t256_body<MODE, false>(p); }
__global__ __launch_bounds__(512, 2) void gemm_bf16_t256_pair(GemmKArgs p) { t256_body<0, true>(p); }

}  // namespace

// tile policy: the 256x256 kernel needs enough tiles to fill 256 CUs and rows to amortise its prologue
// tile policy (both kernels give bit-identical results: same k-order per output element).  The 256x256 kernel is the
// faster one per tile, but it needs enough tiles to fill the 256 CUs and loses to the 128x128 kernel when the last round of
// tiles is mostly empty (measured at M = 1024 / 1536 / 1753, scripts/bench_gemm_small_m.py: 216 tiles 1226 vs 957 TFLOP/s,
// 504 tiles 1384 vs 1069, but 72 tiles 510 vs 680 and 288 tiles 927 vs 993).
// A/B switches are read once per process (not per launch)
static bool env_flag(const char* name) { return getenv(name) != nullptr; }
// rows of tiles of height tm over one problem, or over the two row segments of a pair (each segment starts on a tile boundary)
static long long tile_rows(long long M1, long long M2, int tm) { return (M1 + tm - 1) / tm + (M2 > 0 ? (M2 + tm - 1) / tm : 0); }

static bool use_t256(long long M1, long long M2, int N, int K) {
  static const bool force_t128 = env_flag("DRAG_GEMM_T128");
  if (force_t128 || N < 256 || K < 256) return false;
  const long long M = M1 + M2;
  if (M >= 2048 && M2 == 0) return true;
  if (M < 1024) return false;
  const long long tiles = tile_rows(M1, M2, 256) * ((N + 255) / 256);
  if (tiles < 128) return false;
  if (tiles <= 256) return true;
  const long long rounds = (tiles + 255) / 256;
  return tiles * 10 >= rounds * 256 * 7;              // last-round efficiency >= 0.7
}

// Which kernel a launch the 256x256 rule above does not take should use: 0 = t128, 2 = the 256x256 kernel after all, else
// gemm_bf16_deep<MI, ST, NI> as 100 * (NI == 6) + 10 * MI + ST.  Every choice gives the same bits; the choice is a function of the shape.
//
// Model.  A launch of few tiles is bound by what its busiest CU pulls through the L2 -> LDS path:
//   cost = (workgroups the busiest CU runs) x (tile rows + tile columns)          [per K-step; comparable at equal K]
// and the measurements that calibrate it are taken with COLD weights (COLD=1 in scripts/bench_gemm_small_m.py / bench_gemm_pair.py:
// every launch reads a matrix no recent launch touched) — inside a batch-1 forward 24 GB of weights pass a 256 MB Infinity Cache, so a
// pipeline never sees the hot state in which an isolated benchmark re-reads one matrix (hot, K = 3072 launches tip towards t128's two
// workgroups per CU: (512, 12288, 3072) t128 879 vs 128x192 825; cold 580 vs 839).
//
// Rules, in order (TFLOP/s cold, old choice -> new):
//  * <= 32 rows or < 64 tiles of 64 rows: 32-row tiles, 4-stage ring — (8, 18432, 3072), the AdaLN modulation at batch 8: an HBM
//    stream of the weight, 27 -> 46 hot (5.7 TB/s); (64, 3072, 3072) 38 -> 68.
//  * fallback by tile count: <= 128 tiles of 128x128 -> 64x128 tiles, one workgroup per CU, 4-stage ring; <= 256 -> 64x128, two per CU,
//    3-stage; more -> t128 (two workgroups per CU, 128x128) or 96x128 tiles with a 2-stage ring (two per CU as well) when the model
//    prefers them: the pair (1024 + 512, 3072, 3072) 288 128-row tiles (32 CUs carry two) 696 -> 408 96-row tiles 806 hot;
//    (512, 9216, 3072) 673 -> 820 hot; not (1458, 4304, 1152) 699 vs 497.  Past one round such workgroups come in rounds of 512 and a
//    partly filled last round costs a full one (paired_rounds).
//  * ONE round of one-workgroup-per-CU ring tiles on >= 75 % of the CUs — the cheapest (32 MI) x (128 | 192) tiling that is — replaces the
//    fallback when it cuts the cost by >= 10 %: (1536, 3072, 15360) t128 686 -> 96x192 873 (128x192 821); (512, 12288, 3072) 580 -> 128x192
//    839; the pair (1024 + 512, 3072, 12288) with gate + residual 622 -> 128x192 781; (1024, 3072, 12288) 64x128 511 -> 128x128 681
//    (64x192 632); (729, 4096, 1152) 64x128 393 -> 96x128 487.  Ring depths 2 and 4 of the 192-column tiles were measured too (codes
//    1x2 / 1x4): 4 changes nothing, 2 (two workgroups per CU) loses.
//  * a partly filled second round of 256x256 tiles when even that is cheaper: (1536, 12288, 3072) 96x128 797 / t128 772 -> 908.
// M2 > 0: the rows of a drag_gemm_bf16_pair launch (tile counts are per segment).  *cost_out: the chosen tiling's cost.
static long long paired_rounds(long long tiles) { return tiles <= 256 ? 1 : 2 * ((tiles + 511) / 512); }

static int deep_policy(long long M1, long long M2, int N, int K, long long* cost_out) {
  const long long M = M1 + M2;
  const long long tn = (N + 127) / 128;
  int pick;
  long long cost;                                   // of the 128-column choice, in the units above
  const long long tiles128 = tile_rows(M1, M2, 128) * tn, tiles64 = tile_rows(M1, M2, 64) * tn;
  if (M <= 32 || tiles64 < 64) {                   // 32-row tiles: no MFMA work on rows that do not exist, more workgroups
    if (cost_out) *cost_out = ((tile_rows(M1, M2, 32) * tn + 255) / 256) * (32 + 128);
    return 14;
  }
  if (tiles128 <= 128) { pick = 24; cost = ((tiles64 + 255) / 256) * (64 + 128); }        // <= 256 workgroups of 64 x 128: one per CU, 4-stage ring (96 KiB)
  else if (tiles128 <= 256) { pick = 23; cost = ((tiles64 + 255) / 256) * (64 + 128); }   // <= 512 workgroups: two per CU, 3-stage ring (72 KiB each)
  else { pick = 0; cost = paired_rounds(tiles128) * (128 + 128); }
  static const bool no96 = env_flag("DRAG_GEMM_NO_96");
  if (pick == 0 && !no96) {
    const long long c96 = paired_rounds(tile_rows(M1, M2, 96) * tn) * (96 + 128);
    if (c96 < cost) { pick = 32; cost = c96; }
  }
  // one round of ring tiles (ties: the larger MI at 128 columns)
  static const bool no192 = env_flag("DRAG_GEMM_NO_192");
  long long cbest = 0;
  int best = 0;
  if (M >= 256) {
    for (int mi = 4; mi >= 1; --mi)
      for (int ni = 4; ni <= 6; ni += 2) {
        if (ni == 6 && (N % 192 != 0 || no192)) continue;
        const long long tiles = tile_rows(M1, M2, 32 * mi) * ((N + 32 * ni - 1) / (32 * ni));
        if (tiles > 256 || tiles < 192) continue;
        const long long c = 32 * mi + 32 * ni;
        if (best == 0 || c < cbest) { best = ni == 6 ? 100 + 10 * mi + 3 : (mi >= 3 ? 10 * mi + 3 : 10 * mi + 4); cbest = c; }
      }
  }
  if (best && cbest * 10 <= cost * 9) {
    if (cost_out) *cost_out = cbest;
    return best;
  }
  // a partly filled second round of 256x256 tiles
  static const bool force_t128 = env_flag("DRAG_GEMM_T128");
  if (pick != 24 && pick != 23 && M >= 1024 && N >= 256 && K >= 256 && !force_t128) {
    const long long c256 = ((tile_rows(M1, M2, 256) * ((N + 255) / 256) + 255) / 256) * (256 + 256);
    if (c256 < cost) { pick = 2; cost = c256; }
  }
  if (cost_out) *cost_out = cost;
  return pick;
}

// persistent grid of the 256x256 kernel: one workgroup per CU (128 KiB of LDS each), fewer when there are fewer tiles
static int t256_grid(int ntiles) {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    ncu = n & ~7;                                  // multiple of the 8 XCDs, so a workgroup's tiles stay on its XCD's L2
    if (ncu == 0) ncu = 8;
  }
  static const bool nonpersistent = env_flag("DRAG_GEMM_NONPERSISTENT");
  if (nonpersistent) return ntiles;
  return ntiles < ncu ? ntiles : ncu;
}

static int fill_common(GemmKArgs& k, const void* A, const void* W, void* C, const void* bias, const void* gate,
                       const void* resid, int M, int N, int K, int ldc, int c_rpb, long long c_bs, int ldg, int act,
                       int act_n0, int out_f32) {
  k.A = (const bf16_t*)A; k.W = (const bf16_t*)W; k.C = C;
  k.bias = (const bf16_t*)bias; k.gate = (const bf16_t*)gate; k.resid = (const bf16_t*)resid;
  k.M = M; k.N = N; k.K = K;
  k.cm.rpb = c_rpb > 0 ? c_rpb : M; k.cm.bs = c_bs; k.cm.ld = ldc;
  k.ldg = ldg; k.act = act; k.act_n0 = act_n0; k.out_f32 = out_f32;
  k.a_bytes = 0; k.w_bytes = 0;
  k.ldw = K; k.w_boff = 0; k.split_m1 = 0;
  k.C2 = nullptr; k.ld2 = 0; k.n_split = 0;
  k.seg_tiles_m = 0; k.A2 = nullptr; k.W2 = nullptr; k.Cs2 = nullptr; k.bias2 = nullptr; k.gate2 = nullptr; k.resid2 = nullptr;
  k.M2 = 0; k.ldg2 = 0; k.wide2 = 0; k.am2 = RowMap{1, 0, 0}; k.cm2 = RowMap{1, 0, 0};
  // M tiles per group of the tile walk: the 32 concurrent tiles of an XCD form a group_m x (32 / group_m) super-tile.  4 and 8 tie on
  // the K = 3072 shapes (8 ahead by 1-3 % at N = 3072), 4 is 2-3 % ahead at K >= 12288; 16 / 32 (towards W-stationary) lose 5-10 %
  // everywhere (scripts/bench_gemm_group_m.py, two boxes).  Order only: the bits do not depend on it.
  k.epi_generic = drag_opt(DRAG_OPT_GEMM_EPILOGUE) == 1;
  k.w4_late_state = drag_opt(DRAG_OPT_GEMM_EPILOGUE) == 2;
  // Round 5 (profiles/r05_gemm_tile_walk_fetch_clock.txt: bytes the L2s pull per launch and the clock the launch gets, per setting): the
  // fabric traffic of a launch is A x tiles_n / (32 / g) + W x tiles_m / g and the chip — at its 1400 W socket cap in this kernel — pays for
  // it in clock (g = 1 on (42696, 21504, 3072): 23.4 GB and 1.48 GHz; g = 4: 11.5 GB and 1.77 GHz).  4 | 8 are the two minima; the wide
  // launches (N >= 16384: the single blocks' q|k|v|mlp Linear) run 1.5 % faster on 4.
  k.group_m = drag_opt(DRAG_OPT_GEMM_GROUP_M) > 0 ? drag_opt(DRAG_OPT_GEMM_GROUP_M) : (K >= 8192 || N >= 16384 ? 4 : 8);
  static const bool narrow = env_flag("DRAG_GEMM_NARROW");
  k.wide = !out_f32 && N % 8 == 0 && ldc % 8 == 0 && ((uintptr_t)C & 15) == 0 && (k.cm.rpb >= M || c_bs % 8 == 0) &&
           (!resid || ((uintptr_t)resid & 15) == 0) && (!gate || (((uintptr_t)gate & 15) == 0 && ldg % 8 == 0)) &&
           !narrow;
  k.tiles_m = (M + BM - 1) / BM; k.tiles_n = (N + BN - 1) / BN;
  return k.tiles_m * k.tiles_n;
}

// argument checks + kernel arguments of one row segment
static int gemm_prepare(GemmKArgs& k, const drag_gemm_args* a, bool k_slices = false) {
  DRAG_CHECK(a != nullptr, "drag_gemm_bf16: null args");
  DRAG_CHECK(a->A && a->W && a->C, "drag_gemm_bf16: null operand pointer");
  DRAG_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "drag_gemm_bf16: M, N, K must be positive");
  DRAG_CHECK(a->K % BK == 0, "drag_gemm_bf16: K must be a multiple of 64 (pad the weight)");
  DRAG_CHECK(a->N % 4 == 0, "drag_gemm_bf16: N must be a multiple of 4");
  DRAG_CHECK(a->lda % 8 == 0 && a->ldc % 4 == 0, "drag_gemm_bf16: lda %% 8 and ldc %% 4 required");
  DRAG_CHECK(!(a->gate && !a->resid), "drag_gemm_bf16: gate needs resid");
  DRAG_CHECK(a->act == DRAG_ACT_NONE || a->act == DRAG_ACT_GELU_TANH || a->act == DRAG_ACT_SILU || a->act == DRAG_ACT_QUICK_GELU,
             "drag_gemm_bf16: fused activation must be none, gelu-tanh, silu or quick-gelu (erf GELU: use drag_act_bf16)");
  fill_common(k, a->A, a->W, a->C, a->bias, a->gate, a->resid, a->M, a->N, a->K, a->ldc, a->c_rows_per_batch, a->c_batch_stride, a->ldg,
              a->act, a->act_n0, a->out_f32);
  k.am.rpb = a->a_rows_per_batch > 0 ? a->a_rows_per_batch : a->M;
  k.am.bs = a->a_batch_stride; k.am.ld = a->lda;
  k.cv = ConvMap{1, 1, 1, 1, 64, 1, 0, 0};
  // in-tile offsets are 32-bit and must grow with the row index
  // (k_slices: the row batches of a split-K launch are K slices of the SAME rows — batch stride = the slice width; its tiles never
  //  cross a batch: M % 256 == 0)
  DRAG_CHECK(k_slices || k.am.rpb >= a->M || k.am.bs >= (long long)(k.am.rpb - 1) * k.am.ld,
             "drag_gemm_bf16: a_batch_stride must not be smaller than one batch of rows");
  DRAG_CHECK(((long long)BM * a->lda + a->K) * 2 < (1ll << 30) && (long long)BN * a->K * 2 < (1ll << 31) &&
                 (k_slices || k.am.rpb >= a->M || (k.am.bs - (long long)(k.am.rpb - 1) * k.am.ld) * 2 < (1ll << 30)),
             "drag_gemm_bf16: tile span too large for 32-bit offsets");
  if (a->C2 != nullptr) {
    DRAG_CHECK(a->n_split > 0 && a->n_split < a->N && a->n_split % 256 == 0, "drag_gemm_bf16: n_split must be a multiple of 256 inside (0, N)");
    DRAG_CHECK(!a->out_f32 && !a->gate && !a->resid && a->c_rows_per_batch <= 0, "drag_gemm_bf16: the two-destination form takes dense bf16 outputs without gate / residual");
    DRAG_CHECK(a->ldc2 >= a->N - a->n_split && a->ldc >= a->n_split, "drag_gemm_bf16: ldc / ldc2 too small for their column ranges");
    k.C2 = a->C2; k.ld2 = a->ldc2; k.n_split = a->n_split;
    k.wide = k.wide && a->ldc2 % 8 == 0 && ((uintptr_t)a->C2 & 15) == 0;
    DRAG_CHECK(k.wide || a->ldc2 % 4 == 0, "drag_gemm_bf16: ldc2 % 4 required");
  }
  return 0;
}

// kernel choice for M rows (the sum over the segments of a pair): 2 = t256, 0 = t128, else a gemm_bf16_deep code.
// "gemm_kernel" (drag_set_option, measurement only): 0 = policy, 1 = t128, 2 = t256, 10*MI + ST = gemm_bf16_deep<MI, ST>
static int gemm_choice(long long M1, long long M2, int N, int K, long long* cost_out = nullptr) {
  const int force = drag_opt(DRAG_OPT_GEMM_KERNEL);
  long long cost = 0;
  int choice;
  // 3 = gemm_bf16_w4p (round 5): the same 256 x 256 tiles as four waves x (128 x 128) with the hand-placed K loop — same bits, +1...+5 % on
  // the launches the persistent 8-wave kernel takes (profiles/r05_gemm_w4p_*.log) once a launch has enough tiles (below); one row
  // segment, K a multiple of 128 ("gemm_w4" = 1: never; "gemm_kernel" = 3 forces it wherever it can run)
  const bool w4_ok = M2 == 0 && N >= 256 && K >= 256 && K % 128 == 0;
  if (force >= 10) choice = force;
  else if (force == 3) choice = w4_ok ? 3 : ((N >= 256 && K >= 256) ? 2 : 0);
  else if (force == 2) choice = (N >= 256 && K >= 256) ? 2 : 0;
  else if (force == 1) choice = 0;
  else if (use_t256(M1, M2, N, K)) {
    choice = 2;
    cost = ((tile_rows(M1, M2, 256) * ((N + 255) / 256) + 255) / 256) * (256 + 256);
    // its fixed cost per tile is 7 us against 4.5 and its first tile starts cold: three rounds of tiles, or one with long K loops, amortise
    // that (configs[1]'s two-round (1536, 21 504, 3072) launch lost 2 %: profiles/r05_gemm_w4p_fixed_cost.log)
    const long long tiles4 = tile_rows(M1, 0, 256) * ((N + 255) / 256);
    const int w4 = drag_opt(DRAG_OPT_GEMM_W4);      // 1 never | 2 every launch of the 8-wave kernel it can run | 3 those of >= 256 tiles
    const bool enough = w4 == 2 ? true : w4 == 3 ? tiles4 >= 256 : tiles4 >= 256 && (K >= 8192 || tiles4 >= 768);
    if (w4_ok && w4 != 1 && enough) choice = 3;
  } else choice = deep_policy(M1, M2, N, K, &cost);
  if (cost_out) *cost_out = cost;
  return choice;
}

// one launch over the rows of `a` and, when b != nullptr, of a second segment with the same N, K and epilogue form
// w_total_k > 0: `a` is the stacked form of a split-K launch (gemm_splitk): row batch b of A is K slice b, W's rows are w_total_k wide
static int gemm_launch(const drag_gemm_args* a, const drag_gemm_args* b, void* stream, int w_total_k = 0, const drag_gemm_args* split_second = nullptr) {
  GemmKArgs k, k2;
  if (int rc = gemm_prepare(k, a, w_total_k > 0)) return rc;
  if (w_total_k > 0) {
    k.ldw = w_total_k; k.w_boff = a->K;
    if (split_second) { k.split_m1 = a->a_rows_per_batch - split_second->M; k.A2 = (const bf16_t*)split_second->A; k.W2 = (const bf16_t*)split_second->W; }
  }
  if (b != nullptr) {
    if (int rc = gemm_prepare(k2, b)) return rc;
    k.A2 = k2.A; k.W2 = k2.W; k.Cs2 = k2.C; k.bias2 = k2.bias; k.gate2 = k2.gate; k.resid2 = k2.resid;
    k.M2 = k2.M; k.ldg2 = k2.ldg; k.wide2 = k2.wide; k.am2 = k2.am; k.cm2 = k2.cm;
  }
  const int choice = w_total_k > 0 ? 3 : gemm_choice(a->M, b ? b->M : 0, a->N, a->K);      // (the W offsets of the slices exist in gemm_bf16_w4p only)
  const hipStream_t st_ = (hipStream_t)stream;
  auto tiles_of = [&](int tm) {            // each segment starts on a tile boundary
    k.tiles_m = (a->M + tm - 1) / tm;
    if (b) { k.seg_tiles_m = k.tiles_m; k.tiles_m += (b->M + tm - 1) / tm; }
  };
  if (choice == 3) {            // the 4-wave persistent kernel (one segment, K % 128 == 0: gemm_choice)
    k.tiles_m = (a->M + 255) / 256; k.tiles_n = (a->N + 255) / 256;
    // its own tile walk (profiles/r05_gemm_w4p_group_m.log, interleaved, TFLOP/s at g = 2 | 4 | 8): (42 696, 21 504, 3072) 1468 | 1449 | 1419,
    // (32 768, 9216, 3072) 1473 | 1497 | 1481, (32 768, 3072, 12 288) 1520 | 1528 | 1513, (32 768, 3072, 3072) 1456 | 1423 | 1451
    if (drag_opt(DRAG_OPT_GEMM_GROUP_M) <= 0) k.group_m = a->N >= 16384 ? 2 : (a->N <= 3072 && a->K < 8192 ? 8 : 4);
    constexpr int lds4 = 131072 + 4 * 4096;
    static unsigned long long ready4 = 0;       // one bit per device: the attribute belongs to the device's copy of the kernel
    int dev4 = 0;
    (void)hipGetDevice(&dev4);
    if (!((ready4 >> (dev4 & 63)) & 1ull)) {
      DRAG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_w4p), hipFuncAttributeMaxDynamicSharedMemorySize, lds4) == hipSuccess,
                 "drag_gemm_bf16: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
      ready4 |= 1ull << (dev4 & 63);
    }
    hipLaunchKernelGGL(gemm_bf16_w4p, dim3(t256_grid(k.tiles_m * k.tiles_n)), dim3(256), lds4, st_, k);
  } else if (choice == 2) {
    tiles_of(256); k.tiles_n = (a->N + 255) / 256;
    const dim3 g(t256_grid(k.tiles_m * k.tiles_n));
    if (b) hipLaunchKernelGGL(gemm_bf16_t256_pair, g, dim3(512), 0, st_, k);
    else hipLaunchKernelGGL((gemm_bf16_t256<0>), g, dim3(512), 0, st_, k);
  } else if (choice) {
    const int deep = choice;
    const int ni = deep >= 100 ? 6 : 4, mi = (deep % 100) / 10, st = deep % 10;
    DRAG_CHECK((deep >= 400 && deep <= 405) || ((mi >= 1 && mi <= 4) && st >= 2 && st <= 4), "drag_gemm_bf16: gemm_kernel must be 0, 1, 2, 10 * MI + ST or 100 + 10 * MI + ST of a gemm_bf16_deep<MI, ST, NI> that is built");
    tiles_of(deep >= 400 ? 256 : 32 * mi); k.tiles_n = (a->N + 32 * ni - 1) / (32 * ni);
    const dim3 g(k.tiles_m * k.tiles_n);
#define DRAG_DEEP_LAUNCH(MI_, ST_, NI_)                                                                                    \
  {                                                                                                                        \
    constexpr int lds = ST_ * (32 * MI_ + 32 * NI_) * 128;                                                                 \
    static unsigned long long ready = 0;        /* one bit per device: the attribute belongs to the device's copy of the kernel */ \
    int dev_ = 0;                                                                                                          \
    (void)hipGetDevice(&dev_);                                                                                             \
    if (!((ready >> (dev_ & 63)) & 1ull)) {                                                                                \
      DRAG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_deep<MI_, ST_, NI_>),                       \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess,                      \
                 "drag_gemm_bf16: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");                               \
      ready |= 1ull << (dev_ & 63);                                                                                        \
    }                                                                                                                      \
    hipLaunchKernelGGL((gemm_bf16_deep<MI_, ST_, NI_>), g, dim3(256), lds, st_, k);                                       \
  }                                                                                                                        \
  break
#define DRAG_DEEP(MI_, ST_) case 10 * MI_ + ST_: DRAG_DEEP_LAUNCH(MI_, ST_, 4)
#define DRAG_DEEP6(MI_, ST_) case 100 + 10 * MI_ + ST_: DRAG_DEEP_LAUNCH(MI_, ST_, 6)
#if DRAG_EXP
    if (deep >= 400 && deep <= 405) {            // round-5 experiment: gemm_bf16_w4<variant>
      DRAG_CHECK(a->K % 128 == 0 && a->K >= 256 && b == nullptr, "drag_gemm_bf16: gemm_kernel 400 (gemm_bf16_w4) needs K % 128 == 0, K >= 256, one segment");
      k.tiles_m = (a->M + 255) / 256; k.tiles_n = (a->N + 255) / 256;
      const dim3 g4(k.tiles_m * k.tiles_n);
#define DRAG_W4(V_) case 400 + V_: DRAG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_w4<V_>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072) == hipSuccess, "hipFuncSetAttribute"); \
      hipLaunchKernelGGL((gemm_bf16_w4<V_>), g4, dim3(256), 131072, st_, k); break
      switch (deep) { DRAG_W4(0); DRAG_W4(1); DRAG_W4(2); DRAG_W4(3); DRAG_W4(4); DRAG_W4(5); default: DRAG_CHECK(false, "drag_gemm_bf16: gemm_kernel 400..405"); }
#undef DRAG_W4
      DRAG_LAUNCH_CHECK();
      return 0;
    }
#endif
    switch (deep) {
      DRAG_DEEP(4, 2); DRAG_DEEP(4, 3);
      DRAG_DEEP(3, 2); DRAG_DEEP(3, 3);
      DRAG_DEEP(2, 2); DRAG_DEEP(2, 3); DRAG_DEEP(2, 4);
      DRAG_DEEP(1, 3); DRAG_DEEP(1, 4);
      DRAG_DEEP6(1, 3); DRAG_DEEP6(2, 3); DRAG_DEEP6(3, 3); DRAG_DEEP6(4, 3);
      DRAG_DEEP6(3, 2); DRAG_DEEP6(4, 2); DRAG_DEEP6(3, 4); DRAG_DEEP6(4, 4);
      default: DRAG_CHECK(false, "drag_gemm_bf16: gemm_kernel names a gemm_bf16_deep<MI, ST, NI> that is not built (42 43 32 33 22 23 24 13 14 | 113 123 133 143 132 142 134 144)");
    }
#undef DRAG_DEEP
#undef DRAG_DEEP6
#undef DRAG_DEEP_LAUNCH
  } else {
    tiles_of(BM); k.tiles_n = (a->N + BN - 1) / BN;
    hipLaunchKernelGGL(gemm_bf16_t128<0>, dim3(k.tiles_m * k.tiles_n), dim3(256), 0, st_, k);
  }
  DRAG_LAUNCH_CHECK();
  return 0;
}

// --------------------------------------------------------------------------------------------
// Split-K (round 5): a Linear with few output tiles and a long K — the single blocks' proj_out at batch 1, 1536 x 3072 x 15 360: 72 tiles
// of 256 x 256 on 256 CUs, or 256 tiles of 96 x 192 with half the arithmetic intensity (today's choice: 177 us, L2 -> LDS bound) — runs as
// ONE launch of gemm_bf16_w4p over S stacked K slices ((S M) x N x (K / S), f32 partial products into the caller's workspace: 216 tiles,
// 115 us) and one pass that adds the slices in order and applies the epilogue (splitk_reduce_kernel, the arithmetic of epi_row).  The sum
// of S f32 chains is not the single chain's bits: the one place where the kernel choice changes the last bit of an output — accuracy
// against the f32 oracle is the same (tests) — hence policy only where it pays (below) and "gemm_splitk" = 1 to switch it off.
// The workspace is the caller's (drag_gemm_set_workspace, per device): no allocation on the launch path, safe under graph capture;
// launches that use it must be ordered on one stream.
// --------------------------------------------------------------------------------------------
static void* g_splitk_ws[64];
static long long g_splitk_ws_bytes[64];
extern "C" int drag_gemm_set_workspace(void* ptr, int64_t bytes) {
  int dev = 0;
  DRAG_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "drag_gemm_set_workspace: no current device");
  DRAG_CHECK((ptr == nullptr) == (bytes <= 0) && ((uintptr_t)ptr & 255) == 0, "drag_gemm_set_workspace: a 256-byte aligned device buffer and its size, or (null, 0)");
  g_splitk_ws[dev] = ptr;
  g_splitk_ws_bytes[dev] = ptr ? (long long)bytes : 0;
  return 0;
}

struct SplitReduceArgs {
  const float* part;      // [S][slice_rows][N]; this problem's rows begin at row0 of every slice
  int slice_rows, row0;
  bf16_t* C;
  const bf16_t *bias, *gate, *resid;
  int S, M, N, ldg;
  RowMap cm;
};
// one thread = four consecutive columns of one row: v = bias + sum over the slices in order; then epi_row's forms.  Every load of the thread —
// S partial quads, bias, gate, residual — is issued before the first add (S is a template parameter: with a run-time loop the compiler
// waited for slice s before requesting s + 1: 28 us for the 75 MB of a (1536, 3072) x 3 launch, a third of the HBM rate)
template <int S>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitReduceArgs p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N / 4;
  if (i >= (long long)p.M * n4) return;
  const int m = (int)(i / n4), n = (int)(i - (long long)m * n4) * 4;
  f32x4_t part[S];
#pragma unroll
  for (int s = 0; s < S; ++s) part[s] = __builtin_nontemporal_load((const f32x4_t*)(p.part + ((long long)s * p.slice_rows + p.row0 + m) * p.N + n));
  const long long coff = p.cm.off(m) + n;
  u32x2_t bb = {0u, 0u}, rr = {0u, 0u}, gg = {0u, 0u};
  if (p.bias) bb = *(const u32x2_t*)(p.bias + n);
  if (p.resid) rr = *(const u32x2_t*)(p.resid + coff);
  if (p.gate) gg = *(const u32x2_t*)(p.gate + (long long)(m / p.cm.rpb) * p.ldg + n);
  f32x4_t v = part[0];
#pragma unroll
  for (int s = 1; s < S; ++s) v += part[s];
  if (p.bias) { v[0] += bf_lo(bb[0]); v[1] += bf_hi(bb[0]); v[2] += bf_lo(bb[1]); v[3] += bf_hi(bb[1]); }
  if (p.resid) {
    const float x[4] = {bf_lo(rr[0]), bf_hi(rr[0]), bf_lo(rr[1]), bf_hi(rr[1])};
    if (p.gate) {      // diffusers: x + gate * y with y, gate, x bf16 tensors: y rounded first, the product rounded, then the sum
      const float g[4] = {bf_lo(gg[0]), bf_hi(gg[0]), bf_lo(gg[1]), bf_hi(gg[1])};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = x[r] + rbf(g[r] * rbf(v[r]));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = x[r] + rbf(v[r]);
    }
  }
  u32x2_t o;
  o[0] = pack2bf(v[0], v[1]);
  o[1] = pack2bf(v[2], v[3]);
  *(u32x2_t*)(p.C + coff) = o;
}

// slices of a launch (0: not split).  Policy: at most 96 tiles of 256 x 256 (three eighths of the chip), K >= 12 288, as many slices as fit
// one round of workgroups (<= 8, slices of whole 128-wide K-step pairs, >= 1024 wide); "gemm_splitk": 1 never | n >= 2 that many where valid
static int splitk_slices(const drag_gemm_args* a, const drag_gemm_args* b = nullptr) {
  const int opt = drag_opt(DRAG_OPT_GEMM_SPLITK);
  if (opt == 1 || drag_opt(DRAG_OPT_GEMM_KERNEL) != 0) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || g_splitk_ws[dev] == nullptr) return 0;
  auto fits = [](const drag_gemm_args* x) {
    return x->M % 256 == 0 && x->N % 256 == 0 && !x->out_f32 && !x->C2 && x->act == DRAG_ACT_NONE &&
           !(x->a_rows_per_batch > 0 && x->a_rows_per_batch < x->M) &&          // A is one dense batch
           x->ldc % 4 == 0 && x->ldg % 4 == 0 && !(x->c_rows_per_batch > 0 && x->c_batch_stride % 4);
  };
  if (!fits(a) || (b && (!fits(b) || b->lda != a->lda || b->N != a->N || b->K != a->K))) return 0;
  const long long rows = a->M + (b ? b->M : 0);
  const long long tiles = (rows / 256) * (a->N / 256);
  int best = 0;
  for (int s = 2; s <= 8; ++s) {
    if (a->K % (128 * s) || a->K / s < 1024 || tiles * s > 256) continue;
    if ((long long)s * rows * a->N * 4 > g_splitk_ws_bytes[dev]) continue;
    best = s;
  }
  if (opt >= 2) return (opt <= 8 && a->K % (128 * opt) == 0 && a->K / opt >= 256 && (long long)opt * rows * a->N * 4 <= g_splitk_ws_bytes[dev]) ? opt : 0;
  return (tiles <= 96 && a->K >= 12288) ? best : 0;      // (K = 8192: two slices of 72 tiles lose 3 %: profiles/r05_gemm_splitk_ab.log)
}

static int splitk_reduce(const drag_gemm_args* a, int S, int slice_rows, int row0, const float* part, void* stream) {
  SplitReduceArgs r{};
  r.part = part; r.slice_rows = slice_rows; r.row0 = row0; r.C = (bf16_t*)a->C;
  r.bias = (const bf16_t*)a->bias; r.gate = (const bf16_t*)a->gate; r.resid = (const bf16_t*)a->resid;
  r.S = S; r.M = a->M; r.N = a->N; r.ldg = a->ldg;
  r.cm.rpb = a->c_rows_per_batch > 0 ? a->c_rows_per_batch : a->M; r.cm.bs = a->c_batch_stride; r.cm.ld = a->ldc;
  const long long n = (long long)a->M * (a->N / 4);
  const dim3 g((unsigned)((n + 255) / 256));
  switch (S) {
#define DRAG_RED(S_) case S_: hipLaunchKernelGGL(splitk_reduce_kernel<S_>, g, dim3(256), 0, (hipStream_t)stream, r); break
    DRAG_RED(2); DRAG_RED(3); DRAG_RED(4); DRAG_RED(5); DRAG_RED(6); DRAG_RED(7); DRAG_RED(8);
#undef DRAG_RED
    default: DRAG_CHECK(false, "drag_gemm_bf16: 2 ... 8 K slices");
  }
  DRAG_LAUNCH_CHECK();
  return 0;
}

// b != nullptr: a pair (same N, K, row stride of A): ONE partial launch over S x (M_a + M_b) rows, one reduce pass per problem
static int gemm_splitk(const drag_gemm_args* a, const drag_gemm_args* b, int S, void* stream) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int rows = a->M + (b ? b->M : 0);
  drag_gemm_args v{};
  v.A = a->A; v.W = a->W; v.C = g_splitk_ws[dev];
  v.M = S * rows; v.N = a->N; v.K = a->K / S;
  v.lda = a->lda; v.a_rows_per_batch = rows; v.a_batch_stride = a->K / S;
  v.ldc = a->N; v.c_rows_per_batch = rows; v.c_batch_stride = (long long)rows * a->N;
  v.out_f32 = 1;
  if (int rc = gemm_launch(&v, nullptr, stream, a->K, b)) return rc;
  if (int rc = splitk_reduce(a, S, rows, 0, (const float*)g_splitk_ws[dev], stream)) return rc;
  return b ? splitk_reduce(b, S, rows, a->M, (const float*)g_splitk_ws[dev], stream) : 0;
}

extern "C" int drag_gemm_bf16(const drag_gemm_args* a, void* stream) {
  if (a != nullptr && a->A && a->W && a->C && a->M > 0 && a->N > 0 && a->K > 0) {
    if (const int S = splitk_slices(a)) {
      DRAG_CHECK(!(a->gate && !a->resid), "drag_gemm_bf16: gate needs resid");
      return gemm_splitk(a, nullptr, S, stream);
    }
  }
  return gemm_launch(a, nullptr, stream);
}
extern "C" int drag_gemm_bf16_splitk_slices(const drag_gemm_args* a) { return a ? splitk_slices(a) : 0; }
extern "C" int drag_gemm_bf16_pair_splitk_slices(const drag_gemm_args* a, const drag_gemm_args* b) { return a && b ? splitk_slices(a, b) : 0; }
#if DRAG_EXP
extern "C" int drag_debug_w4_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w4_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// which kernel a launch over M1 (+ M2: a merged pair) rows takes: 2 = the persistent 256x256 kernel, 0 = t128, else
// 100 * (192-column tiles) + 10 * MI + ST of gemm_bf16_deep<MI, ST, NI> — for callers that account launches per kernel
extern "C" int drag_gemm_bf16_choice(int M1, int M2, int N, int K) { return gemm_choice(M1, M2 > 0 ? M2 : 0, N, K); }
// the policy's cost of that launch: tile rounds on the busiest CU x (tile rows + tile columns) — proportional to the bytes the busiest CU
// pulls through its L2 -> LDS path per K-step, which is what bounds a launch of few tiles.  0 under a forced "gemm_kernel".
extern "C" int64_t drag_gemm_bf16_cost(int M1, int M2, int N, int K) {
  long long c = 0;
  gemm_choice(M1, M2 > 0 ? M2 : 0, N, K, &c);
  return (int64_t)c;
}

// Two Linears with their own operands (A, W, bias, gate, residual, output, row maps) but the same N, K, activation and output type as
// ONE launch: the second problem's rows follow the first one's in the tile walk.  Every output element is computed exactly as by
// drag_gemm_bf16 on its own problem (same MFMA chain, same epilogue), so pair(a, b) == gemm(a); gemm(b) bit for bit; what changes is
// the occupancy of launches that are small alone (a double block's 512 text + 1024 image rows at batch 1: BASELINE configs[1]).
// Merged when the cost model above (tile rounds on the busiest CU x (tile rows + tile columns), each launch under the tiling the policy
// gives it) puts the one launch below the two, or level with them while at least one of the two is too small for the persistent kernel
// (a launch of about one round of workgroups pays its ring fill and epilogue in the open; merged, they overlap other workgroups).
// At batch 1 (1024 image + 512 text rows; TFLOP/s over both problems, isolated, scripts/bench_gemm_pair.py): the q|k|v pair
// 878 -> 1282, the attention output pair 559 -> 806, the MLP down-projection pair 667 -> 906; NOT the MLP up-projection (N = 12288:
// one round of 256x256 tiles + 1.5 of t128 alone, 4.5 rounds of t128 merged: 1084 -> 987) and not launches that each fill the chip
// alone (the headline's 4096 text + 32768 image rows: the merged launch idles what the smaller one idles alone).
// "gemm_pair": 0 = this rule, 1 = never merge, 2 = always merge.
extern "C" int drag_gemm_bf16_pair_merges(int M1, int M2, int N, int K) {
  const int opt = drag_opt(DRAG_OPT_GEMM_PAIR);
  if (opt == 1) return 0;
  if (opt == 2) return 1;
  long long c1 = 0, c2 = 0, cm = 0;
  const int k1 = gemm_choice(M1, 0, N, K, &c1);
  const int k2 = gemm_choice(M2, 0, N, K, &c2);
  gemm_choice(M1, M2, N, K, &cm);
  if (cm <= 0) return 0;                           // a forced kernel ("gemm_kernel"): no model
  auto persistent = [](int k) { return k == 2 || k == 3; };      // the 8-wave or the 4-wave 256 x 256 kernel
  return cm < c1 + c2 || (cm == c1 + c2 && (!persistent(k1) || !persistent(k2)));
}

extern "C" int drag_gemm_bf16_pair(const drag_gemm_args* a, const drag_gemm_args* b, void* stream) {
  DRAG_CHECK(a != nullptr && b != nullptr, "drag_gemm_bf16_pair: null args");
  DRAG_CHECK(a->N == b->N && a->K == b->K, "drag_gemm_bf16_pair: the two problems must share N and K");
  DRAG_CHECK(a->act == b->act && a->act_n0 == b->act_n0 && a->out_f32 == b->out_f32, "drag_gemm_bf16_pair: the two problems must share activation and output type");
  DRAG_CHECK(a->C2 == nullptr && b->C2 == nullptr, "drag_gemm_bf16_pair: no two-destination outputs");
  DRAG_CHECK((a->gate != nullptr) == (b->gate != nullptr) && (a->resid != nullptr) == (b->resid != nullptr) && (a->bias != nullptr) == (b->bias != nullptr),
             "drag_gemm_bf16_pair: the two problems must have the same epilogue operands");
  // a pair of few tiles and a long K (the double blocks' ff down-projections at batch 1: (1024 + 512) x 3072 x 12 288) splits as ONE partial
  // launch over both problems' rows + a reduce pass each: 150 -> 112 us alone, configs[1] 103.6 -> 102.7 ms.  (Two separately split launches
  // were 4 % faster than the merged pair alone and 1.2 % of configs[1] slower in the pipeline: profiles/r05_gemm_splitk_pairs_in_configs1.log.)
  if (a->A && a->W && a->C && b->A && b->W && b->C && a->M > 0 && b->M > 0 && a->N > 0 && a->K > 0) {
    if (const int S = splitk_slices(a, b)) {
      DRAG_CHECK(!(a->gate && !a->resid), "drag_gemm_bf16_pair: gate needs resid");
      return gemm_splitk(a, b, S, stream);
    }
  }
  if (!drag_gemm_bf16_pair_merges(a->M, b->M, a->N, a->K)) {
    // two launches: each through drag_gemm_bf16, i.e. under the SAME policy (split-K included) a caller issuing them one by one gets —
    // domain-rag_amd/ops.py gemm_pair does exactly that to account them separately, and the two APIs must give the same bits
    if (int rc = drag_gemm_bf16(a, stream)) return rc;
    return drag_gemm_bf16(b, stream);
  }
  return gemm_launch(a, b, stream);
}

// which kernel drag_conv3x3_bf16 takes for M = B * Ho * Wo output pixels: 2 = the persistent 256x256 kernel, 0 = t128 (the convolutions use
// use_t256 alone, not the ring kernels of gemm_choice) — for callers that account launches per kernel
extern "C" int drag_conv3x3_bf16_choice(int64_t M, int Cout, int Cin) { return use_t256(M, 0, Cout, 9 * Cin) ? 2 : 0; }

extern "C" int drag_conv3x3_bf16(const drag_conv_args* a, void* stream) {
  DRAG_CHECK(a != nullptr && a->x && a->w && a->y, "drag_conv3x3_bf16: null pointer");
  DRAG_CHECK(a->B > 0 && a->Ho > 0 && a->Wo > 0 && a->Hp > 0 && a->Wp > 0, "drag_conv3x3_bf16: bad shape");
  DRAG_CHECK(a->Cin % BK == 0, "drag_conv3x3_bf16: Cin must be a multiple of 64 (zero-pad channels)");
  DRAG_CHECK(a->Cout % 4 == 0 && a->ldy % 4 == 0, "drag_conv3x3_bf16: Cout and ldy must be multiples of 4");
  DRAG_CHECK(a->stride == 1 || a->stride == 2, "drag_conv3x3_bf16: stride 1 or 2");
  DRAG_CHECK(a->act == DRAG_ACT_NONE || a->act == DRAG_ACT_GELU_TANH || a->act == DRAG_ACT_SILU || a->act == DRAG_ACT_QUICK_GELU,
             "drag_conv3x3_bf16: fused activation must be none, gelu-tanh, silu or quick-gelu");
  DRAG_CHECK((a->Ho - 1) * a->stride + a->oy + 2 <= a->Hp - 1 && (a->Wo - 1) * a->stride + a->ox + 2 <= a->Wp - 1,
             "drag_conv3x3_bf16: taps leave the padded input");
  GemmKArgs k;
  const long long M = (long long)a->B * a->Ho * a->Wo;
  DRAG_CHECK(M < (1ll << 31), "drag_conv3x3_bf16: too many output pixels");
  const int grid = fill_common(k, a->x, a->w, a->y, a->bias, nullptr, a->resid, (int)M, a->Cout, 9 * a->Cin, a->ldy, 0, 0,
                               0, a->act, 0, 0);
  k.am.rpb = (int)M; k.am.bs = 0; k.am.ld = a->Cin;
  k.cv = ConvMap{a->Ho, a->Wo, a->Hp, a->Wp, a->Cin, a->stride, a->oy, a->ox};
  DRAG_CHECK(((long long)(BM * a->stride + 3 * a->Wp * 2) * a->Cin) * 2 < (1ll << 30), "drag_conv3x3_bf16: tile span too large");
  if (use_t256(M, 0, a->Cout, 9 * a->Cin)) {
    k.tiles_m = (int)((M + 255) / 256); k.tiles_n = (a->Cout + 255) / 256;
    hipLaunchKernelGGL(gemm_bf16_t256<1>, dim3(t256_grid(k.tiles_m * k.tiles_n)), dim3(512), 0, (hipStream_t)stream, k);
  } else
    hipLaunchKernelGGL(gemm_bf16_t128<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
  DRAG_LAUNCH_CHECK();
  return 0;
}
