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

struct GemmKArgs {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const bf16_t* bias;   // [N] or null
  const bf16_t* gate;   // [batch, ldg] or null:  C = resid + gate[b, n] * (acc + bias)
  const bf16_t* resid;  // same row addressing as C, or null
  int M, N, K;
  RowMap am, cm;
  int ldg;
  int act;
  int act_n0;     // activation applies to columns >= act_n0
  int out_f32;
  unsigned a_bytes, w_bytes;
  int tiles_m, tiles_n;
};

__global__ __launch_bounds__(256, 2) void gemm_bf16_t128(GemmKArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // A0 A1 B0 B1
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
  const int tm = first_m + rem % gsz;
  const int tn = rem / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging addresses: wave w stages 8-row chunks {4w..4w+3} of both tiles ----
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
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
    voffA[i] = (unsigned)((p.am.off(ra) + slot * 8) * 2);
    voffW[i] = (unsigned)(((long long)rw * p.K + slot * 8) * 2);
  }

  auto stage = [&](int buf, int kt) {
    const int soff = kt * (BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      DRAG_LDS char* dA = (DRAG_LDS char*)smem + buf * TILE_BYTES + (w * 4 + i) * 1024;
      DRAG_LDS char* dB = (DRAG_LDS char*)smem + (2 + buf) * TILE_BYTES + (w * 4 + i) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (DRAG_LDS void*)dA, 16, voffA[i], soff, 0, 0);
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
  const int nq = (l >> 4) * 4;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wr * 64 + mi * 16 + (l & 15);
    if (m >= p.M) continue;
    const long long coff = p.cm.off(m);
    const int bidx = m / p.cm.rpb;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wc * 64 + ni * 16 + nq;
      if (n >= p.N) continue;
      float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      if (p.bias) {
        const u32x2_t bb = *(const u32x2_t*)(p.bias + n);
        v[0] += bf2f((bf16_t)(bb[0] & 0xffff)); v[1] += bf2f((bf16_t)(bb[0] >> 16));
        v[2] += bf2f((bf16_t)(bb[1] & 0xffff)); v[3] += bf2f((bf16_t)(bb[1] >> 16));
      }
      if (p.act != DRAG_ACT_NONE && n >= p.act_n0) {
        // torch: y = linear(x) is a bf16 tensor before the activation reads it
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(rbf(v[r]), p.act);
      }
      if (p.gate) {
        // diffusers computes  x = x + gate * y  with y, gate, x bf16 tensors: y is rounded to
        // bf16 first, the product is rounded, then the sum is rounded.
        const u32x2_t gg = *(const u32x2_t*)(p.gate + (long long)bidx * p.ldg + n);
        const u32x2_t rr = *(const u32x2_t*)(p.resid + coff + n);
        const float g[4] = {bf2f((bf16_t)(gg[0] & 0xffff)), bf2f((bf16_t)(gg[0] >> 16)),
                            bf2f((bf16_t)(gg[1] & 0xffff)), bf2f((bf16_t)(gg[1] >> 16))};
        const float x[4] = {bf2f((bf16_t)(rr[0] & 0xffff)), bf2f((bf16_t)(rr[0] >> 16)),
                            bf2f((bf16_t)(rr[1] & 0xffff)), bf2f((bf16_t)(rr[1] >> 16))};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x[r] + rbf(g[r] * rbf(v[r]));
      } else if (p.resid) {
        const u32x2_t rr = *(const u32x2_t*)(p.resid + coff + n);
        v[0] = bf2f((bf16_t)(rr[0] & 0xffff)) + rbf(v[0]); v[1] = bf2f((bf16_t)(rr[0] >> 16)) + rbf(v[1]);
        v[2] = bf2f((bf16_t)(rr[1] & 0xffff)) + rbf(v[2]); v[3] = bf2f((bf16_t)(rr[1] >> 16)) + rbf(v[3]);
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
}

}  // namespace

extern "C" int drag_gemm_bf16(const drag_gemm_args* a, void* stream) {
  DRAG_CHECK(a != nullptr, "drag_gemm_bf16: null args");
  DRAG_CHECK(a->A && a->W && a->C, "drag_gemm_bf16: null operand pointer");
  DRAG_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "drag_gemm_bf16: M, N, K must be positive");
  DRAG_CHECK(a->K % BK == 0, "drag_gemm_bf16: K must be a multiple of 64 (pad the weight)");
  DRAG_CHECK(a->N % 4 == 0, "drag_gemm_bf16: N must be a multiple of 4");
  DRAG_CHECK(a->lda % 8 == 0 && a->ldc % 4 == 0, "drag_gemm_bf16: lda %% 8 and ldc %% 4 required");
  DRAG_CHECK(!(a->gate && !a->resid), "drag_gemm_bf16: gate needs resid");
  GemmKArgs k;
  k.A = (const bf16_t*)a->A; k.W = (const bf16_t*)a->W; k.C = a->C;
  k.bias = (const bf16_t*)a->bias; k.gate = (const bf16_t*)a->gate; k.resid = (const bf16_t*)a->resid;
  k.M = a->M; k.N = a->N; k.K = a->K;
  k.am.rpb = a->a_rows_per_batch > 0 ? a->a_rows_per_batch : a->M;
  k.am.bs = a->a_batch_stride; k.am.ld = a->lda;
  k.cm.rpb = a->c_rows_per_batch > 0 ? a->c_rows_per_batch : a->M;
  k.cm.bs = a->c_batch_stride; k.cm.ld = a->ldc;
  k.ldg = a->ldg; k.act = a->act; k.act_n0 = a->act_n0; k.out_f32 = a->out_f32;
  // byte span of A / W for the buffer descriptors (raw buffers address with 32-bit offsets)
  const long long a_rows_b = (long long)((a->M - 1) / k.am.rpb);
  const long long a_span = (a_rows_b * k.am.bs + (long long)(k.am.rpb - 1) * k.am.ld + a->K) * 2;
  DRAG_CHECK(a_span < (1ll << 31), "drag_gemm_bf16: A span must be < 2 GiB");
  DRAG_CHECK((long long)BN * a->K * 2 < (1ll << 31), "drag_gemm_bf16: K too large");
  k.a_bytes = (unsigned)a_span; k.w_bytes = 0;
  k.tiles_m = (a->M + BM - 1) / BM; k.tiles_n = (a->N + BN - 1) / BN;
  const int grid = k.tiles_m * k.tiles_n;
  hipLaunchKernelGGL(gemm_bf16_t128, dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
  DRAG_LAUNCH_CHECK();
  return 0;
}
