/* oracle/topk.c — CPU restatement of the retrieval scoring/selection order.  TEST INFRASTRUCTURE
 * ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker; never part of the product path.
 *
 * Follows clip_first_stage_retrieval (retrieval/clip100_resnet_style_all_shots.py:396-451):
 *   index = faiss.IndexFlatIP(d); index.add(features); D, I = index.search(query, k)
 * faiss (faiss-cpu 1.10.0 / faiss-gpu 1.7.2, requirements.txt:8-9, un-vendored) computes exact
 * inner products with BLAS sgemm and a heap; neither the accumulation order nor the tie order
 * is specified by faiss.  PARITY UNPINNED against the original faiss (not installable here);
 * this file pins the order the build DEFINES, and the HIP kernel reproduces it bit-for-bit:
 *
 *   score(n, q) = c, with c = 0 and, in this k order,  c = fmaf(corpus[n][k], query[q][k], c):
 *       for blk in 0 .. d/16-1:  for s in 0..3:  for g in 0..3:  k = 16*blk + 4*g + s
 *   ranking: score descending (-0.0 == +0.0, NaN below everything), ties -> lower index first;
 *   k > N is padded with (-FLT_MAX, -1) as faiss does.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float score_one(const float* row, const float* q, int d) {
  float c = 0.0f;
  for (int blk = 0; blk < d / 16; ++blk)
    for (int s = 0; s < 4; ++s)
      for (int g = 0; g < 4; ++g) {
        const int k = 16 * blk + 4 * g + s;
        c = fmaf(row[k], q[k], c);
      }
  return c;
}

static uint32_t okey(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 1u;
  if (u == 0x80000000u) u = 0u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static float okey_inv(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static int cmp_desc(const void* a, const void* b) {
  const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? 1 : (x > y ? -1 : 0);
}

/* scores[q][n] for all pairs (plain, for tests that want the raw scores) */
void oracle_ip_scores(const float* corpus, const float* queries, int64_t N, int d, int Q, float* scores) {
  for (int q = 0; q < Q; ++q)
    for (int64_t n = 0; n < N; ++n) scores[(int64_t)q * N + n] = score_one(corpus + n * d, queries + (int64_t)q * d, d);
}

int oracle_cosine_topk(const float* corpus, const float* queries, int64_t N, int d, int Q, int k, float* out_d,
                       int64_t* out_i) {
  if (d % 16 != 0 || N <= 0 || Q <= 0 || k <= 0) return -1;
  uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
  if (!keys) return -2;
  for (int q = 0; q < Q; ++q) {
    const float* qv = queries + (int64_t)q * d;
    for (int64_t n = 0; n < N; ++n)
      keys[n] = ((uint64_t)okey(score_one(corpus + n * d, qv, d)) << 32) | (uint64_t)(~(uint32_t)n);
    qsort(keys, (size_t)N, sizeof(uint64_t), cmp_desc);
    for (int j = 0; j < k; ++j) {
      if (j < N) {
        out_d[(int64_t)q * k + j] = okey_inv((uint32_t)(keys[j] >> 32));
        out_i[(int64_t)q * k + j] = (int64_t)(~(uint32_t)(keys[j] & 0xffffffffu));
      } else {
        out_d[(int64_t)q * k + j] = -FLT_MAX;
        out_i[(int64_t)q * k + j] = -1;
      }
    }
  }
  free(keys);
  return 0;
}
