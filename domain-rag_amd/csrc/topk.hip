// topk.hip — exact inner-product top-k over a resident fp32 corpus (gfx950).
//
// Replaces faiss.IndexFlatIP.add/search as used by clip_first_stage_retrieval
// (retrieval/clip100_resnet_style_all_shots.py:425-434): scores = corpus · query, the k largest,
// descending.  faiss leaves the accumulation order (BLAS sgemm) and the tie order unspecified;
// this implementation DEFINES both so that results are bit-reproducible (oracle/topk.c is the
// CPU restatement of exactly this order):
//   score(n, q) = fp32 fma chain, c = 0, over k in the order
//                 for blk in 0..d/16:  for s in 0..3:  for g in 0..3:  k = 16*blk + 4*g + s
//   ties -> lower corpus index first.
//
// Scan kernel (HBM-bound: N*d*4 bytes per pass): each wave streams groups of 16 corpus rows
// HBM -> LDS with LDS-DMA (full 256-B row segments, XOR-swizzled), then feeds
// v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain) with 16 rows x 16 queries; the
// reduction over d happens inside the matrix core, no cross-lane shuffles.
// Selection: (score, index) -> unique 64-bit composite key; per-slice radix select in LDS,
// tree-merged, bitonic-sorted.
#include "drag_common.h"
#include <float.h>

namespace {

typedef unsigned long long u64;

// ------------------------------------------------------------------ scan
struct ScanArgs {
  const float* corpus;
  const float* queries;  // [Q, d] (this pass: Q <= 16)
  float* scores;         // [Q, npad]
  long long N, npad;
  int d, Q;
};

__global__ __launch_bounds__(256, 2) void ip_scan_kernel(ScanArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: query image [d/64][16][256 B] | per wave: 2 x 4 KiB staging
  const int nch = p.d / 64;
  char* sQ = smem;
  const int w = wave_id(), l = lane_id();
  char* sA = smem + nch * 4096 + w * 8192;

  // ---- query image (rows >= Q are zero) ----
  for (int i = threadIdx.x; i < 16 * (p.d / 4); i += 256) {
    const int q = i / (p.d / 4), k4 = i - q * (p.d / 4);  // float4 index within the row
    const int c = k4 >> 4, slot = k4 & 15;
    f32x4_t v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (q < p.Q) v = *(const f32x4_t*)(p.queries + (long long)q * p.d + k4 * 4);
    *(f32x4_t*)(sQ + c * 4096 + q * 256 + ((slot ^ q) & 15) * 16) = v;
  }
  __syncthreads();

  const long long ngroups = (p.N + 15) / 16;
  const long long gstride = (long long)gridDim.x * 4;
  const int g = l >> 4, r16 = l & 15;
  const int rd_base = r16 * 256;

  // this lane's DMA role inside a group chunk: instruction i covers rows 4i + (l>>4)
  // physical slot l&15 holds logical slot (l&15) ^ (row&15)
  unsigned lane_off[4];
  int lane_row[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lane_row[i] = 4 * i + (l >> 4);
    lane_off[i] = (unsigned)((((l & 15) ^ (lane_row[i] & 15)) & 15) * 16);
  }

  for (long long grp = (long long)blockIdx.x * 4 + w; grp < ngroups; grp += gstride) {
    const long long row0 = grp * 16;
    const int nvalid = (int)min((long long)16, p.N - row0);
    const float* gbase = p.corpus + row0 * p.d;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, (unsigned)(nvalid * p.d * 4), 0x00020000);
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) voff[i] = (unsigned)(min(lane_row[i], nvalid - 1) * p.d * 4) + lane_off[i];

    auto dma = [&](int buf, int c) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (DRAG_LDS void*)((DRAG_LDS char*)sA + buf * 4096 + i * 1024), 16,
                                                 voff[i], c * 256, 0, 0);
    };

    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    dma(0, 0);
    for (int c = 0; c < nch; ++c) {
      const int buf = c & 1;
      if (c + 1 < nch) {
        dma(buf ^ 1, c + 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const char* a = sA + buf * 4096 + rd_base;
      const char* q = sQ + c * 4096 + rd_base;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int so = (((4 * cc + g) ^ r16) & 15) * 16;
        const f32x4_t av = *(const f32x4_t*)(a + so);
        const f32x4_t qv = *(const f32x4_t*)(q + so);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], qv[s], acc, 0, 0, 0);
      }
      // the next iteration's DMA overwrites `buf^1`... whose reads were consumed by the MFMAs above
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // acc[r] = score[row0 + 4g + r][query r16]
    if (r16 < p.Q) *(f32x4_t*)(p.scores + (long long)r16 * p.npad + row0 + 4 * g) = acc;
  }
}

// ------------------------------------------------------------------ selection
__device__ __forceinline__ unsigned okey(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 1u;  // NaN ranks below every real score
  if (u == 0x80000000u) u = 0u;                     // -0.0 == +0.0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

constexpr int LMAX = 8192;   // elements a selection block holds in LDS
constexpr int KMAX = 2048;

struct SelArgs {
  const float* scores;   // level 1 input  [Q, npad]
  const u64* in;         // level >= 2 input [Q, G_in, k]
  u64* out;              // [Q, G_out, k]
  long long N, npad;
  int slice;             // level 1: rows per block
  int G_in, F;           // level >= 2: groups in, groups merged per block
  int k, kpad;
  int from_scores;
};

__global__ __launch_bounds__(1024) void select_kernel(SelArgs p) {
  __shared__ u64 lst[LMAX];
  __shared__ u64 srt[KMAX];
  __shared__ unsigned hist[256];
  __shared__ unsigned sh_need, sh_cnt, sh_done;
  __shared__ u64 sh_prefix;
  const int tid = threadIdx.x;
  const int gblk = blockIdx.x, q = blockIdx.y;
  int L;
  if (p.from_scores) {
    const long long r0 = (long long)gblk * p.slice;
    L = (int)min((long long)p.slice, p.N - r0);
    const float* s = p.scores + (long long)q * p.npad + r0;
    for (int i = tid; i < L; i += 1024)
      lst[i] = ((u64)okey(s[i]) << 32) | (u64)(~(unsigned)(r0 + i));
  } else {
    const int g0 = gblk * p.F;
    const int ng = min(p.F, p.G_in - g0);
    L = ng * p.k;
    const u64* s = p.in + ((long long)q * p.G_in + g0) * p.k;
    for (int i = tid; i < L; i += 1024) lst[i] = s[i];
  }
  __syncthreads();
  u64* dst = p.out + ((long long)q * gridDim.x + gblk) * p.k;

  if (L <= p.k) {
    for (int i = tid; i < p.kpad; i += 1024) srt[i] = i < L ? lst[i] : 0ull;
  } else {
    // ---- radix select of the k-th largest composite: up to 8 passes of 8 bits from the top.
    // Early exit: once the bin that holds the k-th element contains EXACTLY the number of elements still needed,
    // every element with that prefix is selected and the remaining low bits need not be resolved (with distinct
    // scores this happens after 2-3 passes; only exact score ties ever reach the index bits).
    if (tid == 0) { sh_need = (unsigned)p.k; sh_prefix = 0ull; sh_done = 0u; }
    u64 mask = 0ull;
    for (int pass = 7; pass >= 0; --pass) {
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      if (sh_done) break;
      const u64 prefix = sh_prefix;
      const unsigned need = sh_need;
      for (int i = tid; i < L; i += 1024) {
        const u64 x = lst[i];
        if ((x & mask) == prefix) atomicAdd(&hist[(unsigned)(x >> (8 * pass)) & 255u], 1u);
      }
      __syncthreads();
      // wave 0 scans the 256 bins from the top: lane i owns bins 255-4i .. 252-4i
      if (tid < 64) {
        unsigned h[4], s4 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = hist[255 - 4 * tid - j]; s4 += h[j]; }
        unsigned incl = s4;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned v = __shfl_up(incl, off, 64);
          if (tid >= off) incl += v;
        }
        unsigned run = incl - s4;                       // elements in bins above this lane's first bin
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (run < need && need <= run + h[j]) {
            sh_need = need - run;
            sh_prefix = prefix | ((u64)(255 - 4 * tid - j) << (8 * pass));
            if (h[j] == need - run) sh_done = 1u;       // the whole bin is selected
          }
          run += h[j];
        }
      }
      mask |= (0xffull << (8 * pass));
      __syncthreads();
    }
    const u64 T = sh_prefix;
    if (tid == 0) sh_cnt = 0u;
    __syncthreads();
    for (int i = tid; i < L; i += 1024) {
      const u64 x = lst[i];
      if (x > T) srt[atomicAdd(&sh_cnt, 1u)] = x;
    }
    __syncthreads();
    const int cgt = (int)sh_cnt;  // < k
    for (int i = cgt + tid; i < p.kpad; i += 1024) srt[i] = i < p.k ? T : 0ull;
  }
  __syncthreads();
  // ---- bitonic sort, descending ----
  for (int sz = 2; sz <= p.kpad; sz <<= 1) {
    for (int st = sz >> 1; st > 0; st >>= 1) {
      for (int i = tid; i < p.kpad; i += 1024) {
        const int j = i ^ st;
        if (j > i) {
          const bool desc = (i & sz) == 0;
          const u64 a = srt[i], b = srt[j];
          if (desc ? (a < b) : (a > b)) { srt[i] = b; srt[j] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < p.k; i += 1024) dst[i] = srt[i];
}

__global__ void decode_kernel(const u64* in, float* out_d, long long* out_i, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const u64 x = in[i];
  if (x == 0ull) { out_d[i] = -FLT_MAX; out_i[i] = -1; return; }   // faiss' padding for k > ntotal
  out_d[i] = okey_inv((unsigned)(x >> 32));
  out_i[i] = (long long)(~(unsigned)(x & 0xffffffffull));
}

struct L2Args { float* x; long long rows; int d; };
__global__ __launch_bounds__(256) void l2norm_kernel(L2Args p) {
  const int w = wave_id(), l = lane_id();
  const long long row = (long long)blockIdx.x * 4 + w;
  if (row >= p.rows) return;
  float* xr = p.x + row * p.d;
  // x / x.norm(dim=-1): sequential-in-lane then butterfly; fp32
  float ss = 0.f;
  for (int c = l; c < p.d; c += 64) ss += xr[c] * xr[c];
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  for (int c = l; c < p.d; c += 64) xr[c] = xr[c] / nrm;
}

inline int level1_slice() { return 4096; }

}  // namespace

extern "C" int64_t drag_cosine_topk_workspace_bytes(int64_t N, int32_t Q) {
  if (N <= 0 || Q <= 0) return 0;
  const long long npad = (N + 63) / 64 * 64;
  const long long G1 = (N + level1_slice() - 1) / level1_slice();
  // scores for one pass of <=16 queries + two candidate buffers sized for the worst k
  return 16 * npad * 4 + 2 * 16 * G1 * (long long)KMAX * 8 + 256;
}

// scan only: scores[q, n] = <corpus[n], queries[q]> for Q <= 16 queries, row stride npad = ceil64(N) floats.
// The same kernel, launch geometry and summation order as the first stage of drag_cosine_topk_f32 (bench.py times
// the HBM-bound pass alone through this entry; tests compare it bit for bit with the oracle's score order).
extern "C" int drag_cosine_scores_f32(const float* corpus, const float* queries, int64_t N, int32_t d, int32_t Q,
                                      float* scores, void* stream) {
  DRAG_CHECK(corpus && queries && scores, "drag_cosine_scores_f32: null pointer");
  DRAG_CHECK(N > 0 && Q > 0 && Q <= 16, "drag_cosine_scores_f32: N > 0 and 1 <= Q <= 16 (one scan pass)");
  DRAG_CHECK(d > 0 && d % 64 == 0 && d <= 1024, "drag_cosine_scores_f32: d must be a multiple of 64, <= 1024");
  DRAG_CHECK(N < (1ll << 32) - 1, "drag_cosine_scores_f32: N must fit 32 bits");
  const int lds = (d / 64) * 4096 + 4 * 8192;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)ip_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DRAG_CHECK(e == hipSuccess, "drag_cosine_scores_f32: cannot raise dynamic LDS limit");
  }
  const long long ngroups = (N + 15) / 16;
  const int grid = (int)min((long long)2048, (ngroups + 3) / 4);
  ScanArgs sa;
  sa.corpus = corpus; sa.queries = queries; sa.scores = scores;
  sa.N = N; sa.npad = (N + 63) / 64 * 64; sa.d = d; sa.Q = Q;
  hipLaunchKernelGGL(ip_scan_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, sa);
  DRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int drag_cosine_topk_f32(const float* corpus, const float* queries, int64_t N, int32_t d, int32_t Q,
                                    int32_t k, float* out_d, int64_t* out_i, void* workspace, void* stream) {
  DRAG_CHECK(corpus && queries && out_d && out_i && workspace, "drag_cosine_topk_f32: null pointer");
  DRAG_CHECK(N > 0 && Q > 0, "drag_cosine_topk_f32: N and Q must be positive");
  DRAG_CHECK(d > 0 && d % 64 == 0 && d <= 1024, "drag_cosine_topk_f32: d must be a multiple of 64, <= 1024");
  DRAG_CHECK(k > 0 && k <= KMAX, "drag_cosine_topk_f32: 1 <= k <= 2048");
  DRAG_CHECK(N < (1ll << 32) - 1, "drag_cosine_topk_f32: N must fit 32 bits");
  hipStream_t st = (hipStream_t)stream;
  const long long npad = (N + 63) / 64 * 64;
  const int slice = level1_slice();
  const long long G1 = (N + slice - 1) / slice;
  float* scores = (float*)workspace;
  u64* bufA = (u64*)((char*)workspace + 16 * npad * 4);
  u64* bufB = bufA + 16 * G1 * (long long)KMAX;
  int kpad = 1;
  while (kpad < k) kpad <<= 1;
  const int lds = (d / 64) * 4096 + 4 * 8192;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)ip_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DRAG_CHECK(e == hipSuccess, "drag_cosine_topk_f32: cannot raise dynamic LDS limit");
  }
  const long long ngroups = (N + 15) / 16;
  int grid = (int)min((long long)2048, (ngroups + 3) / 4);

  for (int q0 = 0; q0 < Q; q0 += 16) {
    const int qn = min(16, Q - q0);
    ScanArgs sa;
    sa.corpus = corpus; sa.queries = queries + (long long)q0 * d; sa.scores = scores;
    sa.N = N; sa.npad = npad; sa.d = d; sa.Q = qn;
    hipLaunchKernelGGL(ip_scan_kernel, dim3(grid), dim3(256), lds, st, sa);
    DRAG_LAUNCH_CHECK();
    // level 1
    SelArgs se;
    se.scores = scores; se.in = nullptr; se.out = bufA; se.N = N; se.npad = npad; se.slice = slice;
    se.G_in = 0; se.F = 0; se.k = k; se.kpad = kpad; se.from_scores = 1;
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)G1, qn), dim3(1024), 0, st, se);
    DRAG_LAUNCH_CHECK();
    long long G = G1;
    u64 *cur = bufA, *nxt = bufB;
    const int F = LMAX / k;  // >= 4
    while (G > 1) {
      const long long Gn = (G + F - 1) / F;
      se.scores = nullptr; se.in = cur; se.out = nxt; se.G_in = (int)G; se.F = F; se.from_scores = 0;
      hipLaunchKernelGGL(select_kernel, dim3((unsigned)Gn, qn), dim3(1024), 0, st, se);
      DRAG_LAUNCH_CHECK();
      u64* t = cur; cur = nxt; nxt = t;
      G = Gn;
    }
    const int total = qn * k;
    hipLaunchKernelGGL(decode_kernel, dim3((total + 255) / 256), dim3(256), 0, st, cur, out_d + (long long)q0 * k,
                       (long long*)out_i + (long long)q0 * k, total);
    DRAG_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int drag_l2_normalize_f32(float* x, int64_t rows, int32_t d, void* stream) {
  DRAG_CHECK(x && rows > 0 && d > 0, "drag_l2_normalize_f32: bad args");
  L2Args p{x, rows, d};
  hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  DRAG_LAUNCH_CHECK();
  return 0;
}
